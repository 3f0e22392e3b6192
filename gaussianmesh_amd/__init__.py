"""gaussianmesh_amd -- MI355X-native hot path of IGLICT/GaussianMesh behind the reference's operator API.

  rasterizer.GaussianRasterizationSettings / GaussianRasterizer / NewGaussianRasterizer
  simple_knn.distCUDA2
  deform.SingleObjectDeform (tensor-in deform + the reference attribute names)
All compute runs in csrc/libgmesh_hip.so (hand-written HIP for gfx950) through include/gmesh_hip.h.
"""
import os as _os

# Frames pipelined over several HIP streams only overlap if the runtime has a hardware queue for each: HIP multiplexes streams onto
# FOUR queues by default and streams that share one serialise (four-stream edit loop on MI355X: 4160 frames/s on 4 queues, 5010 on 8;
# tools/queue_env_probe.sh).  The runtime reads GPU_MAX_HW_QUEUES at its FIRST call, not when torch is imported - measured: set after
# `import torch` it still takes effect, after torch.cuda.is_available() it no longer does - so importing this package early is enough.
# An explicit setting of the caller's is left alone.
QUEUES_SET_ON_IMPORT = "GPU_MAX_HW_QUEUES" not in _os.environ
if QUEUES_SET_ON_IMPORT:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, NewGaussianRasterizer  # noqa: F401
from .simple_knn import distCUDA2  # noqa: F401
