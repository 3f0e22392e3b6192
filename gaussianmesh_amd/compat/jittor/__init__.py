"""Torch-backed subset of the `jittor` API used by the reference's python -- NOT Jittor.

Var is torch.Tensor.  Semantics that differ between the two frameworks and are bridged here:
  * Jittor Vars take part in differentiation by default and `.stop_grad()` opts out; here optimizer parameters are
    made leaves with requires_grad=True when they enter a param group (nn.Adam), `.stop_grad()` detaches.
  * `optimizer.backward(loss)` fills param_group["grads"]; the densifier of the reference reads those and edits
    param_group["params"/"m"/"values"] in place (scene/mesh_based_gaussian_model.py:264-278, 411-480).
  * jt.array converts float64 -> float32 and int64 -> int32 (Jittor's auto_convert_64_to_32).
  * jt.max(x, dim) returns the values only; Var.numpy() works on any Var (detaches, copies to host).
  * tensors are created on the HIP device when jt.flags.use_cuda is set and one is present.
"""
import contextlib as _contextlib

import numpy as _np
import torch as _torch

__gaussianmesh_compat__ = True
__version__ = "0+gaussianmesh-compat"

Var = _torch.Tensor
float = float32 = _torch.float32
float64 = _torch.float64
float16 = _torch.float16
int = int32 = _torch.int32
int64 = _torch.int64
int8 = _torch.int8
uint8 = _torch.uint8
bool = _torch.bool


class _Flags:
    def __init__(self):
        object.__setattr__(self, "use_cuda", 0)
        object.__setattr__(self, "auto_convert_64_to_32", 1)

    @property
    def no_grad(self):
        return 0 if _torch.is_grad_enabled() else 1

    def __setattr__(self, k, v):
        if k == "no_grad":
            _torch.set_grad_enabled(not v)
        else:
            object.__setattr__(self, k, v)


flags = _Flags()


def _device():
    return _torch.device("cuda") if (flags.use_cuda and _torch.cuda.is_available()) else _torch.device("cpu")


def _narrow(t):
    if flags.auto_convert_64_to_32:
        if t.dtype == _torch.float64:
            return t.to(_torch.float32)
        if t.dtype == _torch.int64:
            return t.to(_torch.int32)
    return t


def _dt(dtype):
    if dtype is None:
        return None
    if isinstance(dtype, str):
        return {"float": float32, "float32": float32, "float64": float64, "int": int32, "int32": int32, "int64": int64,
                "bool": bool, "uint8": uint8}[dtype]
    return dtype


def array(data, dtype=None):
    if isinstance(data, _torch.Tensor):
        t = data.detach().clone()
    else:
        t = _torch.as_tensor(_np.asarray(data) if not isinstance(data, _np.ndarray) else data)
        t = _narrow(t.clone())
    if dtype is not None:
        t = t.to(_dt(dtype))
    return t.to(_device())


def _shape(shape, more):
    if more:
        return (shape,) + tuple(more)
    if isinstance(shape, (list, tuple, _torch.Size)):
        return tuple(shape)
    return (shape,)


def zeros(shape, *more, dtype=float32):
    return _torch.zeros(_shape(shape, more), dtype=_dt(dtype), device=_device())


def ones(shape, *more, dtype=float32):
    return _torch.ones(_shape(shape, more), dtype=_dt(dtype), device=_device())


def empty(shape, *more, dtype=float32):
    return _torch.empty(_shape(shape, more), dtype=_dt(dtype), device=_device())


def zeros_like(x, dtype=None):
    return _torch.zeros_like(x, dtype=_dt(dtype))


def ones_like(x, dtype=None):
    return _torch.ones_like(x, dtype=_dt(dtype))


def rand(*shape, dtype=float32):
    return _torch.rand(_shape(shape[0], shape[1:]), dtype=_dt(dtype), device=_device())


def randn(*shape, dtype=float32):
    return _torch.randn(_shape(shape[0], shape[1:]), dtype=_dt(dtype), device=_device())


def normal(mean, std, size=None, dtype=float32):
    if isinstance(mean, _torch.Tensor) or isinstance(std, _torch.Tensor):
        return _torch.normal(mean, std)
    return _torch.normal(mean, std, size=_shape(size, ()), dtype=_dt(dtype), device=_device())


def arange(*a, dtype=None):
    return _narrow(_torch.arange(*a, device=_device())) if dtype is None else _torch.arange(*a, dtype=_dt(dtype), device=_device())


def concat(xs, dim=0):
    return _torch.cat(list(xs), dim=dim)


def stack(xs, dim=0):
    return _torch.stack(list(xs), dim=dim)


unsqueeze = _torch.unsqueeze
squeeze = _torch.squeeze
log = _torch.log
exp = _torch.exp
sqrt = _torch.sqrt
abs = _torch.abs
sigmoid = _torch.sigmoid
matmul = _torch.matmul
where = _torch.where
logical_and = _torch.logical_and
logical_or = _torch.logical_or
logical_not = _torch.logical_not
isnan = _torch.isnan


def norm(x, p=2, dim=-1, keepdim=False, keepdims=False, eps=1e-30):
    return _torch.linalg.vector_norm(x, ord=p, dim=dim, keepdim=keepdim or keepdims)


def normalize(x, p=2, dim=1, eps=1e-12):
    return _torch.nn.functional.normalize(x, p=p, dim=dim, eps=eps)


def clamp(x, min_v=None, max_v=None):
    return _torch.clamp(x, min=min_v, max=max_v)


def clamp_min(x, min_v):
    return _torch.clamp(x, min=min_v)


def _reduce(fn):
    def f(x, dim=None, keepdims=False, keepdim=False):
        if dim is None:
            return fn(x)
        r = fn(x, dim=dim, keepdim=keepdims or keepdim)
        return r.values if hasattr(r, "values") else r          # Jittor returns the values only
    return f


max = _reduce(_torch.max)
min = _reduce(_torch.min)
sum = _reduce(_torch.sum)
mean = _reduce(_torch.mean)


def maximum(a, b):
    return _torch.maximum(a, b)


def minimum(a, b):
    return _torch.minimum(a, b)


def cross(a, b, dim=-1):
    return _torch.cross(a, b, dim=dim)


def set_seed(seed):
    _torch.manual_seed(seed)
    _np.random.seed(seed)


def gc():
    return None


def sync_all(device_sync=False):
    if _torch.cuda.is_available():
        _torch.cuda.synchronize()


def save(obj, path):
    _torch.save(obj, path)


def load(path):
    return _torch.load(path, map_location=_device(), weights_only=False)


no_grad = _torch.no_grad
enable_grad = _torch.enable_grad


@_contextlib.contextmanager
def flag_scope(**kw):
    old = {k: getattr(flags, k) for k in kw}
    for k, v in kw.items():
        setattr(flags, k, v)
    try:
        yield
    finally:
        for k, v in old.items():
            setattr(flags, k, v)


def grad(loss, targets, retain_graph=True):
    targets = list(targets)
    gs = _torch.autograd.grad(loss, targets, retain_graph=retain_graph, allow_unused=True)
    return [_torch.zeros_like(t) if g is None else g for g, t in zip(gs, targets)]


class Function:
    """jt.Function: subclass defines execute(self, *args) and grad(self, *grads); instances are callable and
    `Cls.apply(*args)` works.  Non-tensor arguments are passed through; they get no gradient."""

    def __call__(self, *args):
        owner = self

        class _Bridge(_torch.autograd.Function):
            @staticmethod
            def forward(ctx, *a):
                with _torch.no_grad():
                    out = owner.execute(*a)
                ctx.multi = isinstance(out, (tuple, list))
                return tuple(out) if ctx.multi else out

            @staticmethod
            def backward(ctx, *g):
                r = owner.grad(*g)
                r = list(r) if isinstance(r, (tuple, list)) else [r]
                r += [None] * (len(args) - len(r))
                return tuple(x if isinstance(x, _torch.Tensor) else None for x in r[:len(args)])

        return _Bridge.apply(*args)

    @classmethod
    def apply(cls, *args):
        return cls()(*args)


# ---- Var methods Jittor has and torch lacks (or spells differently)
def _stop_grad(self):
    return self.detach()


def _sync(self, *a, **k):
    return self


def _update(self, other):
    """In-place rebind of the storage (shape may change), as the densifier does on Adam state."""
    with _torch.no_grad():
        self.data = other.detach().to(self.dtype)
    return self


def _copy(self):
    return self.clone()


_torch_numpy = _torch.Tensor.numpy


def _numpy(self, *a, **k):
    return _torch_numpy(self.detach().cpu(), *a, **k)


def _var_normalize(self, p=2, dim=1, eps=1e-12):
    return normalize(self, p=p, dim=dim, eps=eps)


# Var methods Jittor code calls on tensors.  These are patched onto torch.Tensor for the whole process - that is what
# compat.install() means (this module is imported by nothing else); compat.uninstall() restores torch's own attributes.
# Note `numpy`: Jittor's Var.numpy() works on device / graph tensors, so the patched one detaches and copies to the host
# where torch would raise.
_PATCHED = getattr(_torch.Tensor, "_gm_compat_originals", None)           # survives a re-import of this module
_first = _PATCHED is None
if _first:
    _PATCHED = {}
    _torch.Tensor._gm_compat_originals = _PATCHED
_torch_numpy = _PATCHED.get("numpy") or _torch_numpy
for _name, _fn in (("stop_grad", _stop_grad), ("sync", _sync), ("update", _update), ("copy", _copy), ("numpy", _numpy),
                   ("normalize", _var_normalize)):
    if _name not in _PATCHED:
        _PATCHED[_name] = getattr(_torch.Tensor, _name, None)
    setattr(_torch.Tensor, _name, _fn)


def _unpatch_tensor():
    for name, orig in _PATCHED.items():
        if orig is None:
            if hasattr(_torch.Tensor, name):
                delattr(_torch.Tensor, name)
        else:
            setattr(_torch.Tensor, name, orig)
    _PATCHED.clear()

from . import nn, linalg, init  # noqa: E402,F401


class _Cudnn:
    class ops:
        @staticmethod
        def cudnn_conv(x, w, stride_h=1, stride_w=1, pad_h=0, pad_w=0, dil_h=1, dil_w=1, groups=1):
            return _torch.nn.functional.conv2d(x, w, None, (stride_h, stride_w), (pad_h, pad_w), (dil_h, dil_w), groups)


cudnn = _Cudnn


def __getattr__(name):
    raise AttributeError("jittor compat subset of gaussianmesh_amd has no '%s' (see gaussianmesh_amd/compat/jittor)" % name)
