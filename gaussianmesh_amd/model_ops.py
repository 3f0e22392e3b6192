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
    def forward(ctx, bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha, mr_weight, joint=None):
        lib = _lib.lib()
        dev = bc.device
        if dev.type != "cuda":
            raise _lib.GmeshError("mesh_activate needs tensors on a HIP (cuda) device; there is no CPU path")
        ins = [_c(t) for t in (bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r)]
        N = ins[0].shape[0]
        f = dict(dtype=torch.float32, device=dev)
        if joint is None:
            xyz = torch.empty((N, 3), **f); scales = torch.empty((N, 3), **f); rots = torch.empty((N, 4), **f); opac = torch.empty((N, 1), **f)
        else:
            # joint = persistent [N + Nb, .] buffers whose tails hold a frozen cloud's rows (renderer.render(bg_gaussian=...)): the kernel
            # writes the leading N rows and the WHOLE buffers are the outputs - what torch.cat([activated, background]) would return,
            # without the four per-iteration copies.  Fresh tensor objects over the same storage every call.
            # CONTRACT: one forward per backward.  The kernel below rewrites the buffers through raw pointers; their version counters
            # are bumped (no launch) so that a backward pass of an EARLIER forward - whose saved means3D / scales / rotations / opacities
            # alias this storage - raises autograd's "modified by an inplace operation" error instead of using the new values.
            xyz, scales, rots, opac = (joint[k].view_as(joint[k]) for k in ("xyz", "scales", "rots", "opac"))
            for t in (xyz, scales, rots, opac):
                torch.autograd.graph.increment_version(t)
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
        return d_bc, d_dist, d_scaling, d_rot, d_op, None, None, None, None, None, None, None, None


def mesh_activate(bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha=4.0, mr_weight=None, joint=None):
    """Returns (xyz, scales, rotations, opacities) and, with mr_weight, a fifth output: mesh_restrict_loss(scales, v1, v2,
    v3, weight=mr_weight) (utils/loss_utils.py:103-108) computed - and differentiated - inside the same two kernels.
    joint: dict of persistent [N + Nb, .] buffers "xyz", "scales", "rots", "opac" (tails = a frozen cloud); the outputs are then the
    whole buffers (the gradient of the tail rows is dropped)."""
    out = _MeshActivate.apply(bc, distance, scaling, rotation, opacity, v1, v2, v3, normal, r, alpha, mr_weight, joint)
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

    # ---- topology changes: the optimizer state follows the parameter rows (scene/mesh_based_gaussian_model.py:411-480)
    def resize(self, keep=None, new_rows=None):
        """Row surgery on every group in ONE pass per tensor: rows selected by `keep` (bool mask [N] or int64 index tensor; None =
        all) stay in their order - parameter and both Adam moments move together, as _prune_optimizer (:425-438) does - and the
        rows of `new_rows[group name]` are appended behind them with ZERO moments, as cat_tensors_to_optimizer (:465-483) does.
        `new_rows` may name the SH group by its own name or give "f_dc" [n,1,3] and "f_rest" [n,15,3].  Every group's parameter
        becomes a new leaf torch.nn.Parameter (the old one, and any .grad on it, is dropped); returns {group name: parameter}."""
        out = {}
        idx = None
        if keep is not None:
            idx = keep.nonzero(as_tuple=False).reshape(-1) if keep.dtype == torch.bool else keep.reshape(-1).long()
        for g in self.param_groups:
            p = g["params"][0]
            ext = None
            if new_rows is not None:
                if g["name"] in new_rows:
                    ext = new_rows[g["name"]]
                elif g["name"] == "f_dc+f_rest" and "f_dc" in new_rows:
                    ext = torch.cat((new_rows["f_dc"], new_rows["f_rest"]), dim=1)
                else:
                    raise KeyError("FusedAdam.resize: no new rows for group %r" % g["name"])
            nk = p.shape[0] if idx is None else idx.numel()
            nn_ = 0 if ext is None else ext.shape[0]
            if ext is not None and tuple(ext.shape[1:]) != tuple(p.shape[1:]):
                raise ValueError("FusedAdam.resize: new rows of group %r have shape %s, parameter rows %s"
                                 % (g["name"], tuple(ext.shape[1:]), tuple(p.shape[1:])))
            with torch.no_grad():
                np_, nm, nv = (torch.empty((nk + nn_,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device) for _ in range(3))
                for dst, src in ((np_, p.detach()), (nm, g["m"][0]), (nv, g["values"][0])):
                    if idx is None:
                        dst[:nk].copy_(src)
                    elif nk:
                        torch.index_select(src, 0, idx, out=dst[:nk])
                if nn_:
                    np_[nk:].copy_(ext.detach())
                    nm[nk:].zero_()
                    nv[nk:].zero_()
            g["params"][0] = torch.nn.Parameter(np_, requires_grad=p.requires_grad)
            g["m"][0], g["values"][0] = nm, nv
            out[g["name"]] = g["params"][0]
        return out

    def prune(self, keep):
        """_prune_optimizer(mask) (:425-438): keep the rows where `keep` is True."""
        return self.resize(keep=keep)

    def append(self, new_rows):
        """cat_tensors_to_optimizer(tensors_dict) (:465-483)."""
        return self.resize(new_rows=new_rows)

    def replace(self, name, tensor):
        """replace_tensor_to_optimizer(tensor, name) (:411-423): new values for one group, both moments reset to zero."""
        for g in self.param_groups:
            if g["name"] == name:
                old = g["params"][0]
                g["params"][0] = torch.nn.Parameter(tensor.detach().clone().contiguous().float(), requires_grad=old.requires_grad)
                g["m"][0] = torch.zeros_like(g["params"][0])
                g["values"][0] = torch.zeros_like(g["params"][0])
                return {name: g["params"][0]}
        raise KeyError(name)

    def rebind(self, name, param):
        """Point group `name` at `param` (same shape and values expected, e.g. after renderer.share_feature_storage moved the SH
        rows into a shared buffer); the moments are kept."""
        for g in self.param_groups:
            if g["name"] == name:
                if tuple(param.shape) != tuple(g["params"][0].shape):
                    raise ValueError("FusedAdam.rebind: shape %s != %s" % (tuple(param.shape), tuple(g["params"][0].shape)))
                g["params"][0] = param
                return
        raise KeyError(name)

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
        # "active" (optional, with "period"): only the first `active` elements of every period have ever had a gradient - the SH
        # coefficients of the degrees the model has switched on so far (Trainer keeps it at 3 (D+1)^2); the rest is left alone
        ACT = arr(C.c_uint32, [int(g.get("active", 0)) for g in live])
        with torch.cuda.device(dev), torch.no_grad():
            _lib.check(lib.gm_adam_step_active(n, P, G, M, V, S, LR, LR2, PER, SPL, ACT, float(self.betas[0]), float(self.betas[1]),
                                               float(self.eps), int(self.n_step), torch.cuda.current_stream(dev).cuda_stream))


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
