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

import numpy as np
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


class Trainer:
    def __init__(self, gaussians, spatial_lr_scale=1.0, **opt):
        o = dict(DEFAULT_OPT); o.update(opt)
        self.opt = SimpleNamespace(**o)
        self.g = gaussians
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

    def update_learning_rate(self):
        lr = self.bc_lr(self.iteration)
        for gr in self.optimizer.param_groups:
            if gr["name"] in ("bc", "distance"):
                gr["lr"] = lr
        return lr

    def step(self, camera, gt_image, background):
        """One iteration; returns (loss tensor, render package).  No host synchronisation besides the rasterizer's
        instance-count read-back."""
        self.iteration += 1
        self.update_learning_rate()
        g = self.g
        if g.screenspace_points.grad is not None:
            g.screenspace_points.grad = None
        pkg = render(camera, g, self.pipe, background)
        loss = photometric_loss(pkg["render"], gt_image, self.opt.lambda_dssim)
        if self.opt.alpha_mrloss:
            mr = pkg.get("mesh_restrict_loss")
            loss = loss + (mr if mr is not None else
                           mesh_restrict_loss(pkg["scale"], pkg["vertex1"], pkg["vertex2"], pkg["vertex3"], weight=self.opt.alpha_mrloss))
        loss.backward()
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return loss.detach(), pkg
