"""distCUDA2: mean squared distance to the 3 nearest neighbours (scene/simple_knn/__init__.py:15-28)."""
import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.lib()
    if points.device.type != "cuda":
        raise _lib.GmeshError("distCUDA2 needs a tensor on a HIP (cuda) device; there is no CPU path")
    pts = points.detach().contiguous().float()
    P = pts.shape[0]
    means = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return means
    with torch.cuda.device(pts.device):
        nbytes = lib.gm_knn_workspace_bytes(P)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=pts.device)
        _lib.check(lib.gm_knn(P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(), nbytes,
                              torch.cuda.current_stream(pts.device).cuda_stream))
    return means
