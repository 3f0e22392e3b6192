"""jt.linalg subset (edittool/__init__.py:204, utils/graphics_utils.py, scene/cameras.py)."""
import torch as _torch


def inv(x):
    return _torch.linalg.inv(x)


def eigh(x):
    w, v = _torch.linalg.eigh(x)
    return w, v


def det(x):
    return _torch.linalg.det(x)
