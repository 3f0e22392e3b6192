"""One optimisation iteration of the mesh-Gaussian training loop, mirroring train_mesh_gaussian.py:80-148 on this
package's operators (SURVEY.md 8f-3 "training loop harness"):

    render(cam, gaussians)  ->  loss = (1 - l) L1 + l (1 - SSIM) + mesh_restrict_loss  ->  backward  ->  Adam step

`Trainer` keeps the per-group learning rates of MeshBasedGaussianModel.training_setup
(scene/mesh_based_gaussian_model.py:242-263) and the exponential position schedule
(utils/general_utils.py:28-62 get_expon_lr_func).  Densification / pruning are host-side bookkeeping of the reference
model class (it runs unchanged on gaussianmesh_amd.compat); this harness is the fixed-topology loop used to time and
test the GPU path end to end.
"""
import math
from types import SimpleNamespace

import torch

from .loss import mesh_restrict_loss, photometric_loss
from .renderer import render


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Learning-rate schedule of the position groups: geometric interpolation from lr_init (step 0) to lr_final
    (step >= max_steps), optionally multiplied by a sine warm-up from lr_delay_mult to 1 over lr_delay_steps; 0 for
    negative steps or when both rates are 0.  Same values as the reference's utils/general_utils.py:28-62
    (same name and arguments so train_mesh_gaussian.py-style code can call it); pinned by tests/golden/schedule.npz."""
    disabled = lr_init == 0.0 and lr_final == 0.0
    log_ratio = 0.0 if disabled else math.log(lr_final / lr_init)

    def rate(step):
        if step < 0 or disabled:
            return 0.0
        progress = min(max(step / max_steps, 0.0), 1.0)
        value = lr_init * math.exp(log_ratio * progress)
        if lr_delay_steps > 0:
            warm = min(max(step / lr_delay_steps, 0.0), 1.0)
            value *= lr_delay_mult + (1.0 - lr_delay_mult) * math.sin(0.5 * math.pi * warm)
        return value
    return rate


# arguments/__init__.py:70-93 OptimizationParams (pinned by tests/golden/schedule.npz)
DEFAULT_OPT = dict(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                   feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, lambda_dssim=0.2, alpha_mrloss=6.0)


class FrozenGaussians:
    """A free-standing, non-trainable cloud (the bg_gaussian argument of render(), gaussian_renderer/__init__.py:100-121):
    activated tensors held as they are."""

    def __init__(self, xyz, scaling, rotation, opacity, features):
        self.get_xyz, self.get_scaling, self.get_rotation, self.get_opacity, self.get_features = xyz, scaling, rotation, opacity, features

    def get_covariance(self, scaling_modifier=1.0):
        q = self.get_rotation
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                         2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        return L @ L.transpose(1, 2)


class Trainer:
    """densify_stats: keep the densification bookkeeping of train_mesh_gaussian.py:119-126 (max_radii2D, accumulated
    view-space gradient norm, visit count) up to date every iteration (one fused kernel).
    sync_free: the rasterizer never waits for the instance count (rasterizer.set_sync_free_training); the status of the
    forward is checked after backward() has been enqueued and an iteration that overflowed its binning buffer is redone.
    bg_gaussian: a FrozenGaussians cloud composited behind the trainable one."""

    def __init__(self, gaussians, spatial_lr_scale=1.0, densify_stats=False, sync_free=False, bg_gaussian=None, **opt):
        o = dict(DEFAULT_OPT); o.update(opt)
        self.opt = SimpleNamespace(**o)
        self.g = gaussians
        self.bg_gaussian = bg_gaussian
        if bg_gaussian is not None and hasattr(gaussians, "_features") and gaussians._features.is_cuda:
            from .renderer import share_feature_storage
            share_feature_storage(gaussians, bg_gaussian)        # before the optimizer captures the parameter
        s = spatial_lr_scale
        # the reference's seven groups; "f_dc" (coefficient 0) and "f_rest" are the two learning rates of the one SH tensor
        groups = [
            {"params": [gaussians._bc], "lr": o["position_lr_init"] * s, "name": "bc"},
            {"params": [gaussians._distance], "lr": o["position_lr_init"] * s, "name": "distance"},
            {"params": [gaussians._features], "lr": o["feature_lr"], "lr_rest": o["feature_lr"] / 20.0, "period": 48, "split": 3,
             "name": "f_dc+f_rest"},
            {"params": [gaussians._opacity], "lr": o["opacity_lr"], "name": "opacity"},
            {"params": [gaussians._scaling], "lr": o["scaling_lr"], "name": "scaling"},
            {"params": [gaussians._rotation], "lr": o["rotation_lr"], "name": "rotation"},
        ]
        from .model_ops import FusedAdam
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        self.bc_lr = get_expon_lr_func(o["position_lr_init"] * s, o["position_lr_final"] * s, lr_delay_mult=o["position_lr_delay_mult"],
                                       max_steps=o["position_lr_max_steps"])
        self.pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False,
                                    mesh_restrict_weight=(o["alpha_mrloss"] or None))
        self.iteration = 0
        self.sync_free = bool(sync_free)
        self.redone = 0                          # iterations repeated because the instance count outgrew the binning buffer
        self.densify_stats = bool(densify_stats)
        if self.densify_stats:
            N, dev = gaussians._bc.shape[0], gaussians._bc.device
            self.max_radii2D = torch.zeros((N,), device=dev)
            self.bc_gradient_accum = torch.zeros((N, 1), device=dev)
            self.denom = torch.zeros((N, 1), device=dev)

    def update_learning_rate(self):
        lr = self.bc_lr(self.iteration)
        for gr in self.optimizer.param_groups:
            if gr["name"] in ("bc", "distance"):
                gr["lr"] = lr
        return lr

    def _forward_backward(self, camera, gt_image, background):
        g = self.g
        if g.screenspace_points.grad is not None:
            g.screenspace_points.grad = None
        pkg = render(camera, g, self.pipe, background, bg_gaussian=self.bg_gaussian)
        loss = photometric_loss(pkg["render"], gt_image, self.opt.lambda_dssim)
        if self.opt.alpha_mrloss:
            mr = pkg.get("mesh_restrict_loss")
            loss = loss + (mr if mr is not None else
                           mesh_restrict_loss(pkg["scale"], pkg["vertex1"], pkg["vertex2"], pkg["vertex3"], weight=self.opt.alpha_mrloss))
        loss.backward()
        return loss, pkg

    def step(self, camera, gt_image, background):
        """One iteration; returns (loss tensor, render package).  Host synchronisation: the rasterizer's instance-count
        read-back, or with sync_free only the (long completed) status words of the forward."""
        from . import rasterizer
        self.iteration += 1
        self.update_learning_rate()
        rasterizer.set_sync_free_training(self.sync_free)
        try:
            loss, pkg = self._forward_backward(camera, gt_image, background)
            if self.sync_free and not rasterizer.verify_sync_free():
                self.redone += 1                 # image was the background: same iteration again with the enlarged buffer
                self.optimizer.zero_grad(set_to_none=True)
                loss, pkg = self._forward_backward(camera, gt_image, background)
                rasterizer.verify_sync_free()
        finally:
            rasterizer.set_sync_free_training(False)
        if self.densify_stats:
            from .model_ops import densify_stats
            N = self.max_radii2D.shape[0]
            densify_stats(pkg["radii"][:N], self.g.screenspace_points.grad, self.max_radii2D, self.bc_gradient_accum, self.denom)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return loss.detach(), pkg
