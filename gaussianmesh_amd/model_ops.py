"""Fused per-Gaussian operators of the training loop around the rasterizer, csrc/gm_train.hip.

  mesh_activate(bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha=4)
      -> (xyz, scales, rotations, opacities): MeshBasedGaussianModel.get_xyz / get_scaling / get_rotation / get_opacity
         (scene/mesh_based_gaussian_model.py:122-152, 172-174) in one kernel, differentiable (one kernel backward).
  FusedAdam(groups, eps, betas)
      jittor.nn.Adam's update rule for all groups in one launch per step (scene/mesh_based_gaussian_model.py:242-263).
"""
import ctypes as C

import torch

from . import _lib


def _c(t):
    return t.detach().contiguous().float()


class _MeshActivate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha, mr_weight):
        lib = _lib.lib()
        dev = bc.device
        if dev.type != "cuda":
            raise _lib.GmeshError("mesh_activate needs tensors on a HIP (cuda) device; there is no CPU path")
        ins = [_c(t) for t in (bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r)]
        N = ins[0].shape[0]
        f = dict(dtype=torch.float32, device=dev)
        xyz = torch.empty((N, 3), **f); scales = torch.empty((N, 3), **f); rots = torch.empty((N, 4), **f); opac = torch.empty((N, 1), **f)
        want_mr = mr_weight is not None
        part = torch.empty(((N + 255) // 256,), **f) if want_mr else None
        with torch.cuda.device(dev):
            _lib.check(lib.gm_mesh_activate_fwd(N, float(alpha), *[t.data_ptr() for t in ins], xyz.data_ptr(), scales.data_ptr(),
                                                rots.data_ptr(), opac.data_ptr(), float(mr_weight or 0.0),
                                                None if part is None else part.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        ctx.save_for_backward(*ins)
        ctx.alpha, ctx.mr_weight = float(alpha), (float(mr_weight) if want_mr else None)
        mr = part.sum() if want_mr else torch.zeros((), **f)
        return xyz, scales, rots, opac, mr

    @staticmethod
    def backward(ctx, d_xyz, d_scales, d_rots, d_opac, d_mr):
        lib = _lib.lib()
        ins = ctx.saved_tensors
        dev = ins[0].device
        N = ins[0].shape[0]
        f = dict(dtype=torch.float32, device=dev)
        d_bc = torch.empty((N, 3), **f); d_dist = torch.empty_like(ins[1]); d_scaling = torch.empty((N, 3), **f)
        d_rot = torch.empty((N, 4), **f); d_op = torch.empty_like(ins[4])
        g = [None if t is None else _c(t) for t in (d_xyz, d_scales, d_rots, d_opac)]
        gm = _c(d_mr).reshape(1) if (ctx.mr_weight is not None and d_mr is not None) else None
        with torch.cuda.device(dev):
            _lib.check(lib.gm_mesh_activate_bwd(N, ctx.alpha, *[t.data_ptr() for t in ins], *[None if t is None else t.data_ptr() for t in g],
                                                d_bc.data_ptr(), d_dist.data_ptr(), d_scaling.data_ptr(), d_rot.data_ptr(), d_op.data_ptr(),
                                                float(ctx.mr_weight or 0.0), None if gm is None else gm.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream))
        return d_bc, d_dist, d_scaling, d_rot, d_op, None, None, None, None, None, None, None


def mesh_activate(bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha=4.0, mr_weight=None):
    """Returns (xyz, scales, rotations, opacities) and, with mr_weight, a fifth output: mesh_restrict_loss(scales, v1, v2,
    v3, weight=mr_weight) (utils/loss_utils.py:103-108) computed - and differentiated - inside the same two kernels."""
    out = _MeshActivate.apply(bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha, mr_weight)
    return out if mr_weight is not None else out[:4]


class FusedAdam:
    """param_groups: list of dicts {"params": [tensor], "lr": float, "name": str, optional "lr_rest", "period", "split"}
    (one tensor per group, as the reference's training_setup builds them).  step() reads .grad of every parameter and
    applies jittor.nn.Adam's rule to all groups in one kernel launch; state ("m", "values" - Jittor's names) lives in the
    group dicts."""

    def __init__(self, param_groups, lr=0.0, eps=1e-8, betas=(0.9, 0.999)):
        self.param_groups = []
        self.eps, self.betas, self.n_step = eps, betas, 0
        for g in param_groups:
            g = dict(g)
            if len(g["params"]) != 1:
                raise ValueError("FusedAdam: one tensor per group")
            p = g["params"][0]
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError("FusedAdam: parameters must be contiguous float32")
            g.setdefault("lr", lr)
            g["m"] = [torch.zeros_like(p)]
            g["values"] = [torch.zeros_like(p)]
            self.param_groups.append(g)
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdam: at most 8 groups per optimizer")

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            p = g["params"][0]
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        lib = _lib.lib()
        live = [g for g in self.param_groups if g["params"][0].grad is not None and g["params"][0].numel() > 0]
        if not live:
            return
        self.n_step += 1
        n = len(live)
        dev = live[0]["params"][0].device
        grads = [g["params"][0].grad.contiguous() for g in live]
        arr = lambda ty, vals: (ty * n)(*vals)
        P = arr(C.c_void_p, [g["params"][0].data_ptr() for g in live])
        G = arr(C.c_void_p, [t.data_ptr() for t in grads])
        M = arr(C.c_void_p, [g["m"][0].data_ptr() for g in live])
        V = arr(C.c_void_p, [g["values"][0].data_ptr() for g in live])
        S = arr(C.c_uint64, [g["params"][0].numel() for g in live])
        LR = arr(C.c_float, [float(g["lr"]) for g in live])
        LR2 = arr(C.c_float, [float(g.get("lr_rest", g["lr"])) for g in live])
        PER = arr(C.c_uint32, [int(g.get("period", 0)) for g in live])
        SPL = arr(C.c_uint32, [int(g.get("split", 0)) for g in live])
        with torch.cuda.device(dev), torch.no_grad():
            _lib.check(lib.gm_adam_step(n, P, G, M, V, S, LR, LR2, PER, SPL, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                        int(self.n_step), torch.cuda.current_stream(dev).cuda_stream))


def densify_stats(radii, viewspace_grad, max_radii2D, grad_accum, denom):
    """gm_densify_stats: the per-iteration densification bookkeeping of train_mesh_gaussian.py:119-126 in one kernel, in
    place on max_radii2D [N], grad_accum [N(,1)], denom [N(,1)] (float32); radii int32 [N], viewspace_grad [N,3]."""
    lib = _lib.lib()
    dev = radii.device
    N = radii.shape[0]
    if N == 0:
        return
    for t in (max_radii2D, grad_accum, denom):
        if not t.is_contiguous() or t.dtype != torch.float32 or t.numel() != N:
            raise ValueError("densify_stats: accumulators must be contiguous float32 of N elements")
    g = viewspace_grad.detach().contiguous().float()
    with torch.cuda.device(dev), torch.no_grad():
        _lib.check(lib.gm_densify_stats(N, radii.contiguous().data_ptr(), g.data_ptr(), max_radii2D.data_ptr(), grad_accum.data_ptr(),
                                        denom.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
