"""gaussianmesh_amd -- MI355X-native hot path of IGLICT/GaussianMesh behind the reference's operator API.

  rasterizer.GaussianRasterizationSettings / GaussianRasterizer / NewGaussianRasterizer
  simple_knn.distCUDA2
  deform.SingleObjectDeform (tensor-in deform + the reference attribute names)
All compute runs in csrc/libgmesh_hip.so (hand-written HIP for gfx950) through include/gmesh_hip.h.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, NewGaussianRasterizer  # noqa: F401
from .simple_knn import distCUDA2  # noqa: F401
