"""gaussianmesh_amd -- MI355X-native hot path of IGLICT/GaussianMesh behind the reference's operator API.

  rasterizer.GaussianRasterizationSettings / GaussianRasterizer / NewGaussianRasterizer
  simple_knn.distCUDA2
  deform.SingleObjectDeform (tensor-in deform + the reference attribute names)
All compute runs in csrc/libgmesh_hip.so (hand-written HIP for gfx950) through include/gmesh_hip.h.
"""
import os as _os

# Importing this package has NO process-wide side effects (until round 4 it exported GPU_MAX_HW_QUEUES=8 at import).  A render loop
# that pipelines frames over several HIP streams calls configure_runtime() FIRST - before the HIP runtime makes its first call.
RUNTIME_CONFIG = {"hw_queues": None, "applied": False, "note": "configure_runtime() not called"}


def configure_runtime(hw_queues=8, ipc_dmabuf=False):
    """Opt-in process settings for pipelined render loops and one-process-per-GPU launches; returns what it did (also kept in
    RUNTIME_CONFIG).  Call it before the first torch.cuda / HIP call of the process - bench.py and the tools do.

    hw_queues: frames pipelined over several HIP streams only overlap if the runtime has a hardware queue for each: HIP multiplexes
      streams onto FOUR queues by default and streams that share one serialise (four-stream edit loop on MI355X: 4160 frames/s on 4
      queues, 5010 on 8; tools/queue_env_probe.sh).  The runtime reads GPU_MAX_HW_QUEUES at its FIRST call (measured: set after
      `import torch` it still takes effect, after torch.cuda.is_available() it no longer does).  A value the caller has exported
      is left alone.  The variable is inherited by subprocesses and read by every HIP library of the process - which is why
      this is a call the integrator makes, not an import side effect.
    ipc_dmabuf: N > 1 on these hosts: RCCL sets up its xGMI peer buffers through HIP IPC handles and the driver supports only the
      dmabuf flavour (HSA_ENABLE_IPC_MODE_LEGACY=0; `hipIpcGetMemHandle: invalid argument` otherwise).
    GM_NO_RUNTIME_CONFIG=1 in the environment turns the call into a no-op (the integrator's launcher owns the environment)."""
    import sys
    out = {"hw_queues": _os.environ.get("GPU_MAX_HW_QUEUES"), "applied": False, "note": ""}
    if _os.environ.get("GM_NO_RUNTIME_CONFIG", "0") == "1":
        out["note"] = "GM_NO_RUNTIME_CONFIG=1: environment left as it is"
    else:
        torch = sys.modules.get("torch")
        late = bool(torch is not None and torch.cuda.is_initialized())
        if hw_queues and "GPU_MAX_HW_QUEUES" not in _os.environ:
            _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
            out["hw_queues"] = str(int(hw_queues)); out["applied"] = not late
            out["note"] = "set GPU_MAX_HW_QUEUES" + (" AFTER the HIP runtime started: no effect on this process" if late else "")
        elif hw_queues:
            out["note"] = "GPU_MAX_HW_QUEUES already set by the caller"
        if ipc_dmabuf:
            _os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out["ipc_mode_legacy"] = _os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    RUNTIME_CONFIG.clear(); RUNTIME_CONFIG.update(out)
    return out


from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, NewGaussianRasterizer  # noqa: F401
from .simple_knn import distCUDA2  # noqa: F401
