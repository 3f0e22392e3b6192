"""One optimisation iteration of the mesh-Gaussian training loop, mirroring train_mesh_gaussian.py:80-148 on this
package's operators (SURVEY.md 8f-3 "training loop harness"):

    render(cam, gaussians)  ->  loss = (1 - l) L1 + l (1 - SSIM) + mesh_restrict_loss  ->  backward  ->  Adam step

`Trainer` keeps the per-group learning rates of MeshBasedGaussianModel.training_setup
(scene/mesh_based_gaussian_model.py:242-263) and the exponential position schedule
(utils/general_utils.py:28-62 get_expon_lr_func).  Topology changes between iterations - the face splits and prunes of
MeshBasedGaussianModel.densify_and_split / densify_and_split_for_init / densify_and_prune / prune_points / reset_opacity
(scene/mesh_based_gaussian_model.py:334-339, 411-563, 596-647) - are Trainer.resize() and the methods built on it: the
parameter rows, BOTH Adam moments, the per-face buffers, the densification statistics, the shared SH storage and the
sync-free capacity all follow the new row set, so the HIP ops run on a cloud whose size changes between two iterations.
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
DEFAULT_OPT = dict(iterations=30000, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                   feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, lambda_dssim=0.2, alpha_mrloss=6.0,
                   densification_interval=200, opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=15000,
                   densify_grad_threshold=0.0002)


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
    sync_free: the rasterizer never waits for the instance count; the status of the forward is checked after backward() has
    been enqueued and an iteration that overflowed its binning buffer is redone.  The capacity guess and the list of unverified
    forwards live in `self.sync_state` (a rasterizer.SyncFreeState owned by THIS trainer and current only inside its step()).
    bg_gaussian: a FrozenGaussians cloud composited behind the trainable one."""

    def __init__(self, gaussians, spatial_lr_scale=1.0, densify_stats=False, sync_free=False, bg_gaussian=None, dense_dc=None, fused_sh_step=True, **opt):
        """dense_dc (default: on for a model on the GPU at SH degree 0): while only degree 0 is active, train coefficient 0 as a dense
        [N,1,3] tensor (MeshBoundGaussians.begin_dense_dc) instead of as 12 bytes of every 192-byte row; oneup_sh_degree() folds it -
        and both Adam moments - back into the rows when the degree is raised.
        fused_sh_step (default on): whenever the [N,16,3] rows are trained (SH degree 2 and 3: 28 000 of the reference's 30 000 iterations,
        train_mesh_gaussian.py:70-71) their Adam step - 48 of a Gaussian's 59 parameters - is applied inside the rasterizer's backward pass (rasterizer.ShStep,
        gm_backward_sh_step) instead of by FusedAdam afterwards: the 192-byte gradient rows are never written or read.  Same update, element for
        element; iterations that take no optimizer step, keep_grads and the python SH route use the ordinary path."""
        o = dict(DEFAULT_OPT); o.update(opt)
        self.opt = SimpleNamespace(**o)
        self.g = gaussians
        self.bg_gaussian = bg_gaussian
        from .renderer import sh_operand
        deg0 = int(getattr(gaussians, "active_sh_degree", 3))
        low = self.dense_width(deg0) is not None and deg0 < int(getattr(gaussians, "max_sh_degree", 3))
        if dense_dc is None:
            dense_dc = hasattr(gaussians, "begin_dense_dc") and gaussians._features.is_cuda
        # round 6: the dense leaf carries the (D+1)^2 ACTIVE coefficients at degrees 0 and 1 ([N,1,3], [N,4,3]), re-made at oneupSHdegree()
        # with both Adam moments carried over (begin_dense_active); from degree 2 on the rows are trained, their step inside the backward
        self.dense_active = bool(dense_dc) and hasattr(gaussians, "begin_dense_dc")
        dense_dc = self.dense_active and low
        if dense_dc:
            import weakref
            gaussians.begin_dense_dc(self.dense_width(int(gaussians.active_sh_degree)))
            gaussians._dense_dc_owner = weakref.ref(self)        # gaussians.oneupSHdegree() then folds through fold_dense_dc()
        elif getattr(gaussians, "_features_dc0", None) is not None:
            # a model another Trainer left in dense mode, handed to one that trains the rows: fold first - this optimizer would
            # otherwise keep stepping a leaf the model drops at its next oneupSHdegree() and SH training would silently stop
            gaussians.end_dense_dc()
            gaussians._dense_dc_owner = None
        if bg_gaussian is not None and hasattr(gaussians, "_features") and gaussians._features.is_cuda:
            from .renderer import share_feature_storage
            share_feature_storage(gaussians, bg_gaussian)        # before the optimizer captures the parameter
        s = spatial_lr_scale
        sh_leaf = sh_operand(gaussians)
        # the reference's seven groups; "f_dc" (coefficient 0) and "f_rest" are the two learning rates of the one SH tensor
        groups = [
            {"params": [gaussians._bc], "lr": o["position_lr_init"] * s, "name": "bc"},
            {"params": [gaussians._distance], "lr": o["position_lr_init"] * s, "name": "distance"},
            {"params": [sh_leaf], "lr": o["feature_lr"], "lr_rest": o["feature_lr"] / 20.0, "period": (3 * sh_leaf.shape[1] if sh_leaf.shape[1] > 1 else 0), "split": 3,
             "name": "f_dc+f_rest"},             # (dense coefficient 0: no period, every element steps with feature_lr)
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
        from .rasterizer import SyncFreeState
        self.sync_state = SyncFreeState(enabled=self.sync_free)
        self.fused_sh_step = bool(fused_sh_step)
        self.sh_steps_fused = 0                  # iterations whose SH step was taken inside the backward pass
        self.keep_grads = False
        self.last_grads = None
        self.adam_active_only = True             # FusedAdam touches only the SH coefficients of the degrees switched on so far (False: all)
        self._sh_degree_seen = int(getattr(gaussians, "active_sh_degree", 3))     # highest SH degree a gradient has been taken at
        self.redone = 0                          # iterations repeated because the instance count outgrew the binning buffer
        self.resizes = 0                         # topology changes applied (resize and everything built on it)
        self.densify_stats = bool(densify_stats)
        if self.densify_stats:
            N, dev = gaussians._bc.shape[0], gaussians._bc.device
            self.max_radii2D = torch.zeros((N,), device=dev)
            self.bc_gradient_accum = torch.zeros((N, 1), device=dev)
            self.denom = torch.zeros((N, 1), device=dev)

    # ------------------------------------------------------------------------------------------------------------------
    # topology changes (scene/mesh_based_gaussian_model.py:411-563, 596-647)
    _PARAM_OF_GROUP = {"bc": "_bc", "distance": "_distance", "f_dc+f_rest": "_features", "opacity": "_opacity", "scaling": "_scaling",
                       "rotation": "_rotation"}
    _ROW_BUFFERS = ("vertex1", "vertex2", "vertex3", "normal", "r", "fid", "vertex_index")

    def resize(self, keep_mask=None, new_rows=None, new_buffers=None):
        """The one primitive under every topology edit.  Rows where keep_mask is True stay, in order (prune_points :440-463 with
        mask = ~keep_mask); new_rows {"bc","distance","f_dc","f_rest" (or "f_dc+f_rest"),"opacity","scaling","rotation"} are
        appended behind them (densification_postfix :485-506) together with new_buffers {"vertex1","vertex2","vertex3","normal",
        "r"[,"fid","vertex_index"]}.  Adam moments of surviving rows move with their rows, appended rows start from zero
        moments (FusedAdam.resize); max_radii2D / bc_gradient_accum / denom of survivors are kept when nothing is appended
        (:452-455) and reset to zero when rows are (:503-505 followed by the prune of the split originals); the SH parameter
        goes back into the storage it shares with the frozen background; screenspace_points is re-made; the sync-free binning
        capacity is scaled with the row count (a too small guess only costs one redone iteration).  Returns the new row count."""
        g = self.g
        n_old = g._bc.shape[0]
        if keep_mask is not None and keep_mask.shape[0] != n_old:
            raise ValueError("Trainer.resize: keep_mask has %d rows, the model %d" % (keep_mask.shape[0], n_old))
        n_new = 0 if new_rows is None else int(next(iter(new_rows.values())).shape[0])
        if n_new and (new_buffers is None or any(k not in new_buffers for k in ("vertex1", "vertex2", "vertex3", "normal", "r"))):
            raise ValueError("Trainer.resize: appended rows need their vertex1/vertex2/vertex3/normal/r buffers")
        idx = None if keep_mask is None else keep_mask.nonzero(as_tuple=False).reshape(-1)
        dense_dc = getattr(g, "_features_dc0", None) is not None
        sh_rows = None
        if dense_dc:
            # the optimizer holds the dense coefficient-0 leaf; the [N,16,3] rows behind it (coefficients 1.. wait for their degree)
            # follow the same row set as a buffer
            if n_new:
                new_rows = dict(new_rows)
                sh_rows = new_rows["f_dc+f_rest"] if "f_dc+f_rest" in new_rows else torch.cat((new_rows.pop("f_dc"), new_rows.pop("f_rest")), dim=1)
                new_rows["f_dc+f_rest"] = sh_rows[:, :g._features_dc0.shape[1]].contiguous()
            with torch.no_grad():
                store = g._features.detach() if idx is None else g._features.detach().index_select(0, idx)
                if n_new:
                    store = torch.cat((store, sh_rows.to(store.dtype)), dim=0)
        params = self.optimizer.resize(keep=idx, new_rows=new_rows if n_new else None)
        for name, attr in self._PARAM_OF_GROUP.items():
            if dense_dc and name == "f_dc+f_rest":
                g._set_dense_dc(params[name])
            else:
                setattr(g, attr, params[name])
        if dense_dc:
            g._features = torch.nn.Parameter(store.contiguous(), requires_grad=g._features.requires_grad)
        with torch.no_grad():
            for b in self._ROW_BUFFERS:
                old = getattr(g, b, None)
                if old is None:
                    continue
                kept = old if idx is None else old.index_select(0, idx)
                if n_new:
                    ext = (new_buffers or {}).get(b)
                    if ext is None:
                        if b in ("fid", "vertex_index"):              # optional bookkeeping the caller did not extend: drop it
                            setattr(g, b, None)
                            continue
                        raise ValueError("Trainer.resize: no new rows for buffer %r" % b)
                    kept = torch.cat((kept, ext.to(kept.dtype).reshape((n_new,) + tuple(kept.shape[1:]))), dim=0)
                setattr(g, b, kept.contiguous())
            if new_buffers is not None and new_buffers.get("v") is not None and getattr(g, "v", None) is not None:
                g.v = torch.cat((g.v, new_buffers["v"].to(g.v.dtype)), dim=0)
        n = g._bc.shape[0]
        g.screenspace_points = torch.zeros((n, 3), dtype=g._bc.dtype, device=g._bc.device, requires_grad=True)
        if self.bg_gaussian is not None and g._features.is_cuda:
            from .renderer import share_feature_storage, sh_operand
            share_feature_storage(g, self.bg_gaussian)
            self.optimizer.rebind("f_dc+f_rest", sh_operand(g))
        if self.densify_stats:
            if n_new:
                dev = g._bc.device
                self.max_radii2D = torch.zeros((n,), device=dev)
                self.bc_gradient_accum = torch.zeros((n, 1), device=dev)
                self.denom = torch.zeros((n, 1), device=dev)
            elif idx is not None:
                self.max_radii2D = self.max_radii2D.index_select(0, idx)
                self.bc_gradient_accum = self.bc_gradient_accum.index_select(0, idx)
                self.denom = self.denom.index_select(0, idx)
        if n_old and n > n_old:
            self.sync_state.scale_capacity(n / n_old)
        self.resizes += 1
        return n

    def prune_points(self, mask):
        """prune_points(mask) (:440-463): remove the rows where mask is True."""
        return self.resize(keep_mask=mask.logical_not())

    def reset_opacity(self):
        """reset_opacity (:334-339): opacity = min(opacity, 0.01), both Adam moments of the group zeroed."""
        g = self.g
        with torch.no_grad():
            o = torch.clamp_max(torch.sigmoid(g._opacity), 0.01)
            g._opacity = self.optimizer.replace("opacity", torch.log(o / (1.0 - o)))["opacity"]

    def densify_and_split(self, selected, N=4):
        """densify_and_split (:508-563) for an explicit selection: every selected row (a Gaussian on its triangle) is replaced by N
        new rows on the sub-triangles of the midpoint split - N = 4: (a,ab,ac), (ab,b,bc), (ac,bc,c), (ab,bc,ac)
        (utils/general_utils.py:133-170); N = 5 adds a copy on the parent triangle (:172-212) - with barycentrics (1/3,1/3,1/3),
        zero normal offset, the parent's scale / (4 * 0.8), rotation, SH rows, opacity, normal, r and fid; the new rows are
        appended, then the selected originals are pruned (:561-562).  Returns the new row count."""
        if N not in (4, 5):
            raise ValueError("densify_and_split: N must be 4 or 5")
        g = self.g
        if selected.dtype != torch.bool:
            raise ValueError("densify_and_split: boolean selection mask expected")
        ns = int(selected.sum().item())
        if ns == 0:
            return g._bc.shape[0]
        with torch.no_grad():
            rep = lambda t: t[selected].unsqueeze(1).repeat_interleave(N, dim=1).reshape((ns * N,) + tuple(t.shape[1:]))
            a, b, c = g.vertex1[selected], g.vertex2[selected], g.vertex3[selected]
            ab, ac, bc = (a + b) / 2, (a + c) / 2, (b + c) / 2
            tri1 = [a, ab, ac, ab] + ([a] if N == 5 else [])
            tri2 = [ab, b, bc, bc] + ([b] if N == 5 else [])
            tri3 = [ac, bc, c, ac] + ([c] if N == 5 else [])
            st = lambda xs: torch.stack(xs, dim=1).reshape(ns * N, 3)
            new_buffers = {"vertex1": st(tri1), "vertex2": st(tri2), "vertex3": st(tri3), "normal": rep(g.normal), "r": rep(g.r)}
            if getattr(g, "fid", None) is not None:
                new_buffers["fid"] = rep(g.fid)
            if getattr(g, "vertex_index", None) is not None and getattr(g, "v", None) is not None:
                # three new mesh vertices per split face, numbered behind the existing ones (general_utils.py:153-168)
                v0 = g.v.shape[0]
                t = torch.arange(ns * 3, device=g.v.device, dtype=g.vertex_index.dtype).reshape(ns, 3) + v0
                vi = g.vertex_index[selected]
                ia, ib, ic = vi[:, 0], vi[:, 1], vi[:, 2]
                rows = [torch.stack((ia, t[:, 0], t[:, 1]), 1), torch.stack((t[:, 0], ib, t[:, 2]), 1), torch.stack((t[:, 1], t[:, 2], ic), 1),
                        torch.stack((t[:, 0], t[:, 2], t[:, 1]), 1)] + ([vi] if N == 5 else [])
                new_buffers["vertex_index"] = torch.stack(rows, dim=1).reshape(ns * N, 3)
                new_buffers["v"] = torch.stack((ab, ac, bc), dim=1).reshape(ns * 3, 3)
            new_rows = {
                "bc": torch.full((ns * N, 3), 1.0 / 3.0, dtype=g._bc.dtype, device=g._bc.device),
                "distance": torch.zeros((ns * N, 1), dtype=g._bc.dtype, device=g._bc.device),
                "f_dc+f_rest": rep(g.get_features.detach()),
                "opacity": rep(g._opacity.detach()),
                "scaling": torch.log(rep(torch.exp(g._scaling.detach())) / (4 * 0.8)),
                "rotation": rep(g._rotation.detach()),
            }
        return self.resize(keep_mask=selected.logical_not(), new_rows=new_rows, new_buffers=new_buffers)

    def densify_and_split_for_init(self, N=4):
        """densify_and_split_for_init (:596-647): split every face (train_mesh_gaussian.py:60-61 repeats it until the model has
        more than 100 000 Gaussians)."""
        return self.densify_and_split(torch.ones((self.g._bc.shape[0],), dtype=torch.bool, device=self.g._bc.device), N)

    def densify_and_prune(self, max_grad, min_opacity=0.005, extent=None, max_screen_size=None, N=4):
        """densify_and_prune (:588-594): rows whose mean view-space gradient norm (bc_gradient_accum / denom, NaN -> 0) reaches
        max_grad are split; as in the reference, min_opacity / extent / max_screen_size are accepted and unused."""
        if not self.densify_stats:
            raise ValueError("densify_and_prune needs Trainer(densify_stats=True)")
        grads = self.bc_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        return self.densify_and_split(grads.reshape(-1) >= max_grad, N)

    def update_learning_rate(self):
        lr = self.bc_lr(self.iteration)
        for gr in self.optimizer.param_groups:
            if gr["name"] in ("bc", "distance"):
                gr["lr"] = lr
        return lr

    def _sh_step(self):
        """The SH group's step for rasterizer.ShStep, or None when this iteration's SH step is FusedAdam's: while a dense leaf is trained
        (degrees 0 and 1), on the CPU, without an [N,16,3] leaf, on the python SH route."""
        g = self.g
        if not self.fused_sh_step or self.keep_grads or getattr(g, "_features_dc0", None) is not None:
            return None
        if getattr(self.pipe, "convert_SHs_python", False):
            return None
        grp = next((gr for gr in self.optimizer.param_groups if gr["name"] == "f_dc+f_rest"), None)
        if grp is None:
            return None
        p = grp["params"][0]
        if not p.is_cuda or p.dim() != 3 or p.shape[1] != 16 or not p.requires_grad:
            return None
        from .rasterizer import ShStep
        return ShStep(p.detach(), grp["m"][0], grp["values"][0], grp["lr"], grp.get("lr_rest", grp["lr"]), self.optimizer.betas, self.optimizer.eps,
                      self.optimizer.n_step + 1)

    def _forward_backward(self, camera, gt_image, background, sh_step=False):
        g = self.g
        if g.screenspace_points.grad is not None:
            g.screenspace_points.grad = None
        ss = self._sh_step() if sh_step else None
        import contextlib
        with (ss if ss is not None else contextlib.nullcontext()):
            pkg = render(camera, g, self.pipe, background, bg_gaussian=self.bg_gaussian)
            loss = photometric_loss(pkg["render"], gt_image, self.opt.lambda_dssim)
            if self.opt.alpha_mrloss:
                mr = pkg.get("mesh_restrict_loss")
                loss = loss + (mr if mr is not None else
                               mesh_restrict_loss(pkg["scale"], pkg["vertex1"], pkg["vertex2"], pkg["vertex3"], weight=self.opt.alpha_mrloss))
            loss.backward()
        self._sh_fused = ss is not None and ss.applied
        return loss, pkg

    def schedule(self, iteration, white_background=False):
        """What train_mesh_gaussian.py:66-148 does besides render / loss / backward at `iteration` (1-based), from the
        OptimizationParams in self.opt (arguments/__init__.py:70-93): {"oneup": raise the SH degree before rendering (:70-71),
        "stats": keep the densification statistics (:119-124), "densify": densify_and_prune(densify_grad_threshold, 0.005, extent,
        size_threshold, 5) after the statistics - and NO optimizer step in that iteration (:126-128, update_flag :139),
        "size_threshold", "reset_opacity" (:129-130), "optimizer_step" (:137-139)}."""
        o = self.opt
        before_until = iteration < o.densify_until_iter
        densify = before_until and iteration > o.densify_from_iter and iteration % o.densification_interval == 0
        return {"oneup": iteration % 1000 == 0, "stats": before_until, "densify": densify,
                "size_threshold": 20 if iteration > o.opacity_reset_interval else None,
                "reset_opacity": before_until and (iteration % o.opacity_reset_interval == 0 or
                                                   (white_background and iteration == o.densify_from_iter)),
                "optimizer_step": iteration < o.iterations and not densify}

    def train_iteration(self, camera, gt_image, background, white_background=False, extent=None):
        """One iteration of the reference's loop INCLUDING its schedule (self.schedule): SH degree ramp, densification statistics,
        densify_and_prune every densification_interval iterations past densify_from_iter (N = 5, the optimizer step of that
        iteration skipped, as the reference's update_flag does), opacity reset.  Returns (loss, render package, plan) with
        plan["rows"] = the row count after the iteration."""
        plan = self.schedule(self.iteration + 1, white_background)
        if plan["densify"] and not self.densify_stats:           # (short loops that never reach a densification iteration need no statistics)
            raise ValueError("train_iteration: iteration %d densifies (the reference's schedule) and this Trainer keeps no densification "
                             "statistics: build it with densify_stats=True" % (self.iteration + 1))
        if plan["oneup"] and hasattr(self.g, "oneupSHdegree"):
            self.oneup_sh_degree()
        loss, pkg = self.step(camera, gt_image, background, stats=plan["stats"],
                              densify=(self.opt.densify_grad_threshold, 0.005, extent, plan["size_threshold"], 5) if plan["densify"] else None,
                              optimizer_step=plan["optimizer_step"])
        if plan["reset_opacity"]:
            self.reset_opacity()
        plan["rows"] = self.g._bc.shape[0]
        return loss, pkg, plan

    def fold_dense_dc(self):
        """The dense coefficient-0 leaf (dense_dc) goes back into the [N,16,3] rows together with BOTH Adam moments (the other
        coefficients have never had a gradient: their moments are zero, exactly what the reference's Adam holds for them), and the SH
        group steps the rows again.  Called by the model's oneupSHdegree() (train_mesh_gaussian.py:70-71)."""
        g = self.g
        if getattr(g, "_features_dc0", None) is None:
            return
        grp = next(gr for gr in self.optimizer.param_groups if gr["name"] == "f_dc+f_rest")
        m_dc, v_dc = grp["m"][0], grp["values"][0]
        K = m_dc.shape[1]
        shared = getattr(g, "_features_with_bg", None)
        g.end_dense_dc()                                         # rows current again (in place)
        if self.bg_gaussian is not None and shared is not None:
            from .renderer import share_feature_storage
            share_feature_storage(g, self.bg_gaussian)           # the ROWS share their storage with the background from here on
        with torch.no_grad():
            m = torch.zeros_like(g._features); v = torch.zeros_like(g._features)
            m[:, :K].copy_(m_dc); v[:, :K].copy_(v_dc)
        grp["params"][0], grp["m"][0], grp["values"][0] = g._features, m, v
        grp["period"] = 3 * g._features.shape[1]

    @staticmethod
    def dense_width(degree):
        """Coefficients the dense leaf carries at an SH degree, or None where the [N,16,3] rows are trained: (D+1)^2 = 1 and 4 at degrees
        0 and 1 (12- and 48-byte rows instead of 192).  At degree 2 a dense leaf would be 9 coefficients = 108 bytes (12 with whole
        granules): measured at 2 M Gaussians / 4K no faster than the rows (3.54 against 3.58 ms per iteration) - from there on the rows'
        Adam step runs inside the backward pass over the active granules (fused_sh_step: 3.3 ms)."""
        return {0: 1, 1: 4}.get(int(degree))

    def begin_dense_active(self):
        """After oneupSHdegree() folded the dense leaf into the rows: while the NEW degree is still below the model's full one, the
        (D+1)^2 active coefficients become the dense leaf again - [N,4,3] at degree 1, [N,9,3] at degree 2 - with both Adam moments of those
        coefficients carried over from the rows' (the new coefficients' are zero: they have never had a gradient).  The rasterizer takes the
        leaf as `shs` with M = (D+1)^2; at the full degree the rows themselves are trained (and stepped inside the backward, fused_sh_step)."""
        g = self.g
        if (getattr(g, "_features_dc0", None) is not None or int(g.active_sh_degree) >= int(g.max_sh_degree) or not g._features.is_cuda or
                self.dense_width(int(g.active_sh_degree)) is None):
            return
        grp = next(gr for gr in self.optimizer.param_groups if gr["name"] == "f_dc+f_rest")
        K = self.dense_width(int(g.active_sh_degree))
        m_rows, v_rows = grp["m"][0], grp["values"][0]
        leaf = g.begin_dense_dc(K)
        if self.bg_gaussian is not None:
            from .renderer import share_feature_storage, sh_operand
            share_feature_storage(g, self.bg_gaussian)           # the LEAF shares its storage with the background's leading K coefficients
            leaf = sh_operand(g)
        grp["params"][0] = leaf
        grp["m"][0], grp["values"][0] = m_rows[:, :K].contiguous(), v_rows[:, :K].contiguous()
        grp["period"] = 3 * K if K > 1 else 0
        grp.pop("active", None)

    def oneup_sh_degree(self):
        self.g.oneupSHdegree()

    def step(self, camera, gt_image, background, stats=True, densify=None, optimizer_step=True):
        """One iteration; returns (loss tensor, render package).  Host synchronisation: the rasterizer's instance-count
        read-back, or with sync_free only the (long completed) status words of the forward.
        stats=False: leave the densification statistics alone (iterations past densify_until_iter).
        densify: None, or the argument tuple of densify_and_prune, applied after this iteration's statistics.
        optimizer_step=False: gradients are taken (and dropped) without an Adam step - what the reference does in an iteration
        that changed the topology."""
        self.iteration += 1
        self.update_learning_rate()
        st = self.sync_state
        st.enabled = self.sync_free
        will_step = optimizer_step and densify is None            # (an SH step taken inside the backward cannot be taken back)
        with st:
            loss, pkg = self._forward_backward(camera, gt_image, background, will_step)
            attempts = 0
            while self.sync_free and not st.verify():
                # the image was the background and the render gradients zero: the same iteration again, with the buffer
                # verify() just enlarged; the third attempt takes the exact-count path, which cannot overflow
                attempts += 1
                self.redone += 1
                if attempts >= 2:
                    st.enabled = False
                self.optimizer.zero_grad(set_to_none=True)
                loss, pkg = self._forward_backward(camera, gt_image, background, will_step)     # (the refused attempt's backward stepped nothing: device-side check)
                if attempts >= 2:
                    break
        if self.densify_stats and stats:
            from .model_ops import densify_stats
            N = self.max_radii2D.shape[0]
            densify_stats(pkg["radii"][:N], self._viewspace_grad(pkg), self.max_radii2D, self.bc_gradient_accum, self.denom)
        if self.keep_grads:                      # diagnostics / tests: the gradients this step consumed, by group name
            self.last_grads = {gr["name"]: gr["params"][0].grad for gr in self.optimizer.param_groups}
            self.last_grads["viewspace"] = self._viewspace_grad(pkg)[:self.g._bc.shape[0]]
        self._sh_degree_seen = max(self._sh_degree_seen, int(getattr(self.g, "active_sh_degree", 3)))
        if densify is not None:
            self.densify_and_prune(*densify)
        if optimizer_step and densify is None:
            # SH coefficients above the highest degree a gradient was ever taken at have g = m = v = 0: Adam leaves them as they are,
            # FusedAdam does not even read them ("active", gm_adam_step_active) - 45 of a Gaussian's 60 parameters at degree 0
            for gr in self.optimizer.param_groups:
                if gr.get("period") == 48 and gr["params"][0].shape[1] == 16:
                    gr["active"] = 3 * (self._sh_degree_seen + 1) ** 2 if (self.adam_active_only and self._sh_degree_seen < 3) else 0
            self.optimizer.step()
            self.sh_steps_fused += 1 if getattr(self, "_sh_fused", False) else 0
        self.optimizer.zero_grad(set_to_none=True)
        return loss.detach(), pkg

    def _viewspace_grad(self, pkg):
        """viewspace_point_tensor.grad (train_mesh_gaussian.py:119-124): of the render package's probe when it is a leaf (the joint
        [fg; bg] probe of renderer._joint_buffers), else of the model's own (the probe went through a torch.cat)."""
        vp = pkg["viewspace_points"]
        return vp.grad if (vp.is_leaf and vp.grad is not None) else self.g.screenspace_points.grad

    def copy_state_from(self, other):
        """Make this trainer's optimisation state equal to `other`'s (same row count): parameter values (in place - views and
        shared storage stay), both Adam moments, step counters, densification statistics.  For tests and A/B runs that compare
        two configurations of one iteration from the SAME state."""
        with torch.no_grad():
            for ga, gb in zip(self.optimizer.param_groups, other.optimizer.param_groups):
                ga["params"][0].copy_(gb["params"][0]); ga["m"][0].copy_(gb["m"][0]); ga["values"][0].copy_(gb["values"][0])
                ga["lr"] = gb["lr"]
            if self.densify_stats and other.densify_stats:
                self.max_radii2D.copy_(other.max_radii2D); self.bc_gradient_accum.copy_(other.bc_gradient_accum); self.denom.copy_(other.denom)
        self.optimizer.n_step = other.optimizer.n_step
        self.iteration = other.iteration
