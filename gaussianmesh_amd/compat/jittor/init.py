"""jt.init subset."""
import torch as _torch


def eye(shape, dtype=_torch.float32):
    from . import _device, _dt
    n = shape if isinstance(shape, int) else shape[0]
    m = n if isinstance(shape, int) or len(shape) < 2 else shape[1]
    return _torch.eye(n, m, dtype=_dt(dtype), device=_device())
