"""Render glue: the step immediately before / after the rasterizer in the reference.

Mirrors gaussian_renderer/__init__.py:
  render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, bg_gaussian=None)   (:26-143)
  bg_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mesh_gaussians=None) (:146-260)
and the edit tool's ObjectVisualTool.render_gaussian (edittool/__init__.py:400-475) as render_deformed().
Same argument names, same returned dict keys; tensors are torch tensors on a HIP device.  `pc` is any object with the
reference model's properties (get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_covariance(),
active_sh_degree, max_sh_degree, screenspace_points [+ vertex1..3 for the mesh-bound model]); MeshBoundGaussians below
is a minimal torch holder of the mesh-bound parameterisation (scene/mesh_based_gaussian_model.py:122-174).
"""
import math

import torch

from .deform import sh_colors
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, NewGaussianRasterizer, new_work_hint


_WORK_HINTS = True


def set_work_hints(on):
    """render() / bg_render() keep a work-hint buffer (rasterizer.new_work_hint, 4 bytes per 16-px tile) on every camera object they
    are called with: the forward blend then dispatches the tiles that were expensive the last time THIS camera was rendered
    first.  A scheduling aid only - images and gradients do not depend on it.  set_work_hints(False) switches it off."""
    global _WORK_HINTS
    _WORK_HINTS = bool(on)


def camera_work_hint(cam, device):
    """The work-hint buffer riding on a camera object (created on first use; None when hints are off or the object takes no
    attributes)."""
    if not _WORK_HINTS or torch.device(device).type != "cuda":
        return None
    W, H = int(cam.image_width), int(cam.image_height)
    h = getattr(cam, "_gm_work_hint", None)
    if h is None or h[0] != (W, H) or h[1].device != torch.device(device):
        try:
            h = ((W, H), new_work_hint(W, H, device))
            cam._gm_work_hint = h
        except (AttributeError, TypeError):
            return None
    return h[1]


def _settings(cam, bg_color, scaling_modifier, sh_degree, debug):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree,
        campos=cam.camera_center, prefiltered=False, debug=debug, work_hint=camera_work_hint(cam, cam.world_view_transform.device))


def strip_symmetric(cov):
    """[N,3,3] -> [N,6] (xx,xy,xz,yy,yz,zz), utils/general_utils.py:64-72."""
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
          -0.5900435899266435)


def eval_sh_torch(deg, feats, dirs):
    """Real spherical harmonics up to degree 3 as differentiable torch ops: feats [N,K,3] (coefficient-major, as the
    rasterizer takes them), dirs [N,3] unit vectors -> [N,3].  Same basis and constants as the kernels (csrc/gm_sh.h;
    the reference evaluates this in python with utils/sh_utils.py:57-112 eval_sh when pipe.convert_SHs_python is set)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = _SH_C0 * feats[:, 0]
    if deg > 0:
        r = r - _SH_C1 * y * feats[:, 1] + _SH_C1 * z * feats[:, 2] - _SH_C1 * x * feats[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + _SH_C2[0] * xy * feats[:, 4] + _SH_C2[1] * yz * feats[:, 5] + _SH_C2[2] * (2.0 * zz - xx - yy) * feats[:, 6] +
             _SH_C2[3] * xz * feats[:, 7] + _SH_C2[4] * (xx - yy) * feats[:, 8])
        if deg > 2:
            r = (r + _SH_C3[0] * y * (3.0 * xx - yy) * feats[:, 9] + _SH_C3[1] * xy * z * feats[:, 10] +
                 _SH_C3[2] * y * (4.0 * zz - xx - yy) * feats[:, 11] + _SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * feats[:, 12] +
                 _SH_C3[4] * x * (4.0 * zz - xx - yy) * feats[:, 13] + _SH_C3[5] * z * (xx - yy) * feats[:, 14] +
                 _SH_C3[6] * x * (xx - 3.0 * yy) * feats[:, 15])
    return r


def _python_sh_colors(pc, cam, xyz, feats):
    """pipe.convert_SHs_python: colours from SH on the python side of the op (gaussian_renderer/__init__.py:84-92),
    differentiable with respect to the features AND the positions (through the view direction) like the reference's.
    When no gradient can be asked for (inference), one forward-only HIP kernel does the same."""
    if torch.is_grad_enabled() and (feats.requires_grad or xyz.requires_grad):
        d = xyz - cam.camera_center.reshape(1, 3)
        d = d / d.norm(dim=1, keepdim=True)
        return torch.clamp_min(eval_sh_torch(pc.active_sh_degree, feats, d) + 0.5, 0.0)
    return sh_colors(xyz.detach(), cam.camera_center, feats.detach(), rot=None, deg=pc.active_sh_degree)


class _SharedRows(torch.autograd.Function):
    """out = whole buffer, of which `head` is the leading rows (same storage): what torch.cat([head, tail]) would return,
    without moving a byte.  The gradient of `head` is the leading rows of the buffer's gradient."""

    @staticmethod
    def forward(ctx, head, whole):
        ctx.n = head.shape[0]
        ctx.set_materialize_grads(False)         # no gradient for the buffer (the SH step was taken inside the rasterizer's backward,
        return whole.view_as(whole)              # rasterizer.ShStep) must stay NO gradient for the head: zeros would make the optimizer step it

    @staticmethod
    def backward(ctx, grad):
        return (None if grad is None else grad[:ctx.n]), None


def sh_operand(pc):
    """The tensor a model hands to the rasterizer as `shs`: its full [N,16,3] rows, or - MeshBoundGaussians.begin_dense_dc(), while only
    degree 0 is active - the dense [N,1,3] tensor of coefficient 0 (M = 1: the operator reads (D+1)^2 = 1 coefficient of M)."""
    dc0 = getattr(pc, "_features_dc0", None)
    return pc._features if dc0 is None else dc0


def share_feature_storage(pc, bg_gaussian):
    """One-time setup for render(..., bg_gaussian=...) in a loop: the model's SH operand ([N,16,3] `_features`, or the dense [N,1,3]
    coefficient-0 tensor of begin_dense_dc()) becomes a view of the leading rows of one buffer that also holds the (frozen) background's
    rows, so that the per-iteration torch.cat([fg, bg]) of 192-byte rows (0.58 GB at 3 M Gaussians) disappears.  The parameter stays a
    leaf; optimizers that update it in place (FusedAdam, torch.optim.*) keep the buffer current.  Call before creating the optimizer."""
    f = sh_operand(pc)
    b = bg_gaussian.get_features[:, :f.shape[1]]
    whole = torch.empty((f.shape[0] + b.shape[0],) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
    whole[:f.shape[0]].copy_(f.detach())
    whole[f.shape[0]:].copy_(b.detach())
    leaf = torch.nn.Parameter(whole[:f.shape[0]], requires_grad=f.requires_grad)
    if getattr(pc, "_features_dc0", None) is not None:
        pc._set_dense_dc(leaf) if hasattr(pc, "_set_dense_dc") else setattr(pc, "_features_dc0", leaf)
    else:
        pc._features = leaf
    pc._features_with_bg = (whole, bg_gaussian)


def _features_with_background(pc, bg_gaussian, shs):
    shared = getattr(pc, "_features_with_bg", None)
    if shared is not None and shared[1] is bg_gaussian and shs is sh_operand(pc) and shs.data_ptr() == shared[0].data_ptr():
        return _SharedRows.apply(shs, shared[0])
    return torch.cat([shs, bg_gaussian.get_features[:, :shs.shape[1]]], dim=0)


def _joint_buffers(pc, bg_gaussian):
    """Persistent [N + Nb, .] buffers for render(..., bg_gaussian=...) on the fused route: the tails hold the frozen cloud's positions /
    scales / rotations / opacities (written once), the fused activation writes the leading N rows every iteration, and "ss" is the
    screen-space probe of all N + Nb rows (a leaf: its .grad is the reference's viewspace_point_tensor.grad).  Replaces five
    torch.cat per iteration (rocprofv3, C5: eight CatArrayBatchedCopy launches, ~0.13 ms per iteration).  Rebuilt when the row count
    changes (topology edits) or another background is given."""
    N = pc._bc.shape[0]
    jb = getattr(pc, "_joint_buffers", None)
    if jb is not None and jb["N"] == N and jb["bg"] is bg_gaussian and jb["xyz"].device == pc._bc.device:
        return jb
    Nb = bg_gaussian.get_xyz.shape[0]
    f = dict(dtype=torch.float32, device=pc._bc.device)
    jb = {"N": N, "bg": bg_gaussian, "xyz": torch.empty((N + Nb, 3), **f), "scales": torch.empty((N + Nb, 3), **f), "rots": torch.empty((N + Nb, 4), **f),
          "opac": torch.empty((N + Nb, 1), **f), "ss": torch.zeros((N + Nb, 3), requires_grad=True, **f)}
    with torch.no_grad():
        jb["xyz"][N:].copy_(bg_gaussian.get_xyz); jb["scales"][N:].copy_(bg_gaussian.get_scaling)
        jb["rots"][N:].copy_(bg_gaussian.get_rotation); jb["opac"][N:].copy_(bg_gaussian.get_opacity.reshape(-1, 1))
    import weakref
    ref = weakref.ref(pc)

    def to_model(grad, n=N):                     # pc.screenspace_points.grad keeps meaning what it means without a background cloud
        m = ref()
        if m is not None and m.screenspace_points.shape[0] == n:
            m.screenspace_points.grad = grad[:n]
        return grad
    jb["ss"].register_hook(to_model)
    pc._joint_buffers = jb
    return jb


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, bg_gaussian=None):
    """gaussian_renderer/__init__.py:26-143.  Returns {"render", "viewspace_points", "visibility_filter", "radii",
    "vertex1", "vertex2", "vertex3", "scale"}.
    With bg_gaussian on the fused route (a MeshBoundGaussians on the GPU, pipe.compute_cov3D_python off) the activations and the screen-space
    probe live in PERSISTENT joint [fg; bg] buffers (_joint_buffers; `pc.joint_buffers = False` opts out and concatenates per call):
    one forward per backward - the next render() of the same model rewrites the storage the previous call's graph saved (its backward
    then raises autograd's in-place error, model_ops._MeshActivate) and resets the probe's .grad; pkg["scale"] is a view of that storage."""
    fused = hasattr(pc, "activated") and not pipe.compute_cov3D_python and pc._bc.is_cuda
    joint = _joint_buffers(pc, bg_gaussian) if (fused and bg_gaussian is not None and getattr(pc, "joint_buffers", True)) else None
    if joint is not None:
        screenspace_points = joint["ss"]
        screenspace_points.grad = None
    else:
        screenspace_points = pc.screenspace_points
        if bg_gaussian is not None:
            screenspace_points = torch.cat([screenspace_points, torch.zeros_like(bg_gaussian.get_xyz)], dim=0)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, bg_color, scaling_modifier, pc.active_sh_degree, pipe.debug))
    mrloss = None
    if fused:                                    # one kernel for the four activations instead of ~15 elementwise ops
        mr_w = getattr(pipe, "mesh_restrict_weight", None)
        act = pc.activated(mr_w, joint)
        means3D, scales, rotations, opacity = act[:4]
        mrloss = act[4] if mr_w is not None else None
        means2D, cov3D_precomp = screenspace_points, None
    else:
        means3D, means2D, opacity = pc.get_xyz, screenspace_points, pc.get_opacity
        scales = rotations = cov3D_precomp = None
        if pipe.compute_cov3D_python:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            colors_precomp = _python_sh_colors(pc, viewpoint_camera, means3D if joint is None else means3D[:joint["N"]], pc.get_features)
        else:
            shs = sh_operand(pc) if hasattr(pc, "_features") else pc.get_features
    else:
        colors_precomp = override_color
    if bg_gaussian is not None and joint is not None:     # means3D / opacity / scales / rotations already ARE [fg; bg] (the joint buffers)
        if shs is not None:
            shs = _features_with_background(pc, bg_gaussian, shs)
        else:
            bgc = sh_colors(bg_gaussian.get_xyz, viewpoint_camera.camera_center, bg_gaussian.get_features, rot=None, deg=3)
            colors_precomp = torch.cat([colors_precomp, bgc], dim=0)
    elif bg_gaussian is not None:    # background cloud appended (:100-121: the reference concatenates precomputed covariances)
        means3D = torch.cat([means3D, bg_gaussian.get_xyz], dim=0)
        opacity = torch.cat([opacity, bg_gaussian.get_opacity], dim=0)
        if cov3D_precomp is not None:
            bgc3 = bg_gaussian.get_covariance(1.0)
            cov3D_precomp = torch.cat([cov3D_precomp, strip_symmetric(bgc3) if bgc3.dim() == 3 else bgc3], dim=0)
        else:                        # same covariances, built inside the op from the concatenated scales / rotations
            scales = torch.cat([scales, bg_gaussian.get_scaling], dim=0)
            rotations = torch.cat([rotations, bg_gaussian.get_rotation], dim=0)
        if shs is not None:
            shs = _features_with_background(pc, bg_gaussian, shs)
        else:
            bgc = sh_colors(bg_gaussian.get_xyz, viewpoint_camera.camera_center, bg_gaussian.get_features, rot=None, deg=3)
            colors_precomp = torch.cat([colors_precomp, bgc], dim=0)
    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
                                       scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    out = {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
           "vertex1": getattr(pc, "vertex1", None), "vertex2": getattr(pc, "vertex2", None),
           "vertex3": getattr(pc, "vertex3", None), "scale": scales if (bg_gaussian is None or scales is None) else scales[:pc.screenspace_points.shape[0]]}  # None on the compute_cov3D_python route (:143)
    if mrloss is not None:                       # extra key (pipe.mesh_restrict_weight set): the loss term of train_mesh_gaussian.py:93
        out["mesh_restrict_loss"] = mrloss
    return out


def bg_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mesh_gaussians=None):
    """gaussian_renderer/__init__.py:146-260: background model trained with the (frozen) mesh Gaussians composited in."""
    screenspace_points = pc.screenspace_points
    if mesh_gaussians is not None:
        screenspace_points = torch.cat([screenspace_points, mesh_gaussians.screenspace_points], dim=0)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, bg_color, scaling_modifier, pc.active_sh_degree, pipe.debug))
    fused = hasattr(pc, "activated") and not pipe.compute_cov3D_python and pc._bc.is_cuda
    mrloss = None
    if fused:                                    # one kernel for the four activations instead of ~15 elementwise ops
        mr_w = getattr(pipe, "mesh_restrict_weight", None)
        act = pc.activated(mr_w)
        means3D, scales, rotations, opacity = act[:4]
        mrloss = act[4] if mr_w is not None else None
        means2D, cov3D_precomp = screenspace_points, None
    else:
        means3D, means2D, opacity = pc.get_xyz, screenspace_points, pc.get_opacity
        scales = rotations = cov3D_precomp = None
        if pipe.compute_cov3D_python:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            colors_precomp = _python_sh_colors(pc, viewpoint_camera, means3D, pc.get_features)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    if mesh_gaussians is not None:   # .stop_grad() in the reference (:229-233)
        means3D = torch.cat([means3D, mesh_gaussians.get_xyz.detach()], dim=0)
        scales = torch.cat([scales, mesh_gaussians.get_scaling.detach()], dim=0)
        rotations = torch.cat([rotations, mesh_gaussians.get_rotation.detach()], dim=0)
        shs = torch.cat([shs, mesh_gaussians.get_features.detach()], dim=0)
        opacity = torch.cat([opacity, mesh_gaussians.get_opacity.detach()], dim=0)
    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
                                       scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


def render_deformed(viewpoint_camera, objects, bg_color=None):
    """ObjectVisualTool.render_gaussian (edittool/__init__.py:400-475): concatenate the deformed objects
    (SingleObjectDeform instances after .deform()), colours from SH with the deformation-rotated view direction,
    NewGaussianRasterizer with colors_precomp + cov3D_precomp on a white background."""
    dev = objects[0].gaussian_deform_pos.device
    bg = torch.ones(3, device=dev) if bg_color is None else bg_color
    cat = (lambda xs: xs[0] if len(xs) == 1 else torch.cat(xs, dim=0))
    means3D = cat([o.gaussian_deform_pos for o in objects])
    shs = cat([o.gaussian_feature for o in objects])
    rot = cat([o.gaussian_deform_rot for o in objects])
    cov = cat([o.gaussian_deform_cov for o in objects])
    opacity = cat([o.gaussian_o for o in objects])
    colors_precomp = sh_colors(means3D, viewpoint_camera.camera_center, shs, rot=rot, deg=3)
    rasterizer = NewGaussianRasterizer(_settings(viewpoint_camera, bg, 1, 3, False))
    image, _ = rasterizer(means3D=means3D, means2D=torch.zeros_like(means3D), shs=None, colors_precomp=colors_precomp,
                          opacities=opacity, scales=None, rotations=None, cov3D_precomp=strip_symmetric(cov))
    return image


class Camera:
    """Holder with the attribute names of scene/cameras.py:16-50 (Camera) / edittool/camera_utils.py (GSCamera)."""

    def __init__(self, cam_dict, device):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
        self.image_width, self.image_height = cam_dict["W"], cam_dict["H"]
        self.FoVx, self.FoVy = cam_dict["fovx"], cam_dict["fovy"]
        self.world_view_transform = t(cam_dict["view"])
        self.full_proj_transform = t(cam_dict["proj"])
        self.camera_center = t(cam_dict["campos"])


class MeshBoundGaussians(torch.nn.Module):
    """Mesh-bound parameterisation of scene/mesh_based_gaussian_model.py (parameters and activations only; no
    densification / optimiser / PLY):  get_xyz = softmax(bc).(v1,v2,v3) + 4 r (sigmoid(d) - 0.5) n   (:138-152).
    The SH coefficients are ONE parameter `_features` [N,16,3] (`_features_dc` / `_features_rest` are views of it, so
    get_features needs no per-iteration concatenation of the 192-byte rows; the optimizer gives coefficient 0 and the
    rest their own learning rates, train.py).  activated() evaluates all four activations in one fused HIP kernel."""
    alpha_distance = 4

    def __init__(self, bc, distance, features_dc, features_rest, scaling, rotation, opacity, vertex1, vertex2, vertex3, normal, r,
                 sh_degree=3, fid=None, vertex_index=None, v=None):
        """fid [N] (face of the ORIGINAL proxy mesh a Gaussian descends from), vertex_index [N,3] and v [Vm,3] (the refined
        mesh the face splits build, :596-647) are optional bookkeeping: only the topology edits of train.Trainer move them."""
        super().__init__()
        P = torch.nn.Parameter
        self._bc, self._distance = P(bc), P(distance)
        self._features = P(torch.cat((features_dc, features_rest), dim=1).contiguous())
        self._features_dc0 = None                # begin_dense_dc(): the dense [N,1,3] leaf that is trained while only degree 0 is active
                                                 # (never a registered parameter: parameters() / state_dict() keep a fresh model's keys)
        self._scaling, self._rotation, self._opacity = P(scaling), P(rotation), P(opacity)
        for n, t in dict(vertex1=vertex1, vertex2=vertex2, vertex3=vertex3, normal=normal, r=r).items():
            self.register_buffer(n, t)
        self.register_buffer("fid", fid if fid is not None else torch.arange(bc.shape[0], device=bc.device, dtype=torch.int32))
        self.register_buffer("vertex_index", vertex_index)
        self.register_buffer("v", v)
        self.max_sh_degree = self.active_sh_degree = sh_degree
        self.screenspace_points = torch.zeros_like(vertex1, requires_grad=True)

    @property
    def get_number(self):
        return self._bc.shape[0]

    def oneupSHdegree(self):
        owner = getattr(self, "_dense_dc_owner", None)
        owner = owner() if owner is not None else None
        if self._features_dc0 is not None:       # a Trainer that trains the dense leaf folds it back together with its Adam moments
            if owner is not None:
                owner.fold_dense_dc()
            else:
                self.end_dense_dc()
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
        if owner is not None and getattr(owner, "dense_active", False):
            owner.begin_dense_active()           # ... and trains the (D+1)^2 active coefficients of the NEW degree densely (below the full degree)

    def begin_dense_dc(self, K=None):
        """K (default (active_sh_degree + 1)^2; round 6: 1, 4 or 9 - round 5 knew K = 1 only): the leading K coefficients of every row as
        a dense [N,K,3] leaf; the text below is round 5's for K = 1.
        While the model renders at SH degree 0 (the first 1000 iterations of train_mesh_gaussian.py:70-71) only coefficient 0 of every
        192-byte SH row is read, differentiated and stepped - 12 bytes in whole memory sectors of four strided arrays (parameter, gradient,
        both Adam moments).  This moves coefficient 0 into a dense [N,1,3] leaf that the rasterizer takes as `shs` with M = 1
        (tools/sh_dc_probe.py, 2 M Gaussians at 4K: preprocess forward 0.163 -> 0.119 ms, backward 0.203 -> 0.086, the Adam step of the SH
        group 0.353 -> 0.026).  Coefficient 0 of `_features` is stale until end_dense_dc() / oneupSHdegree() folds the leaf back;
        get_features / _features_dc always show the current values."""
        K = (self.active_sh_degree + 1) ** 2 if K is None else int(K)
        if K < (self.active_sh_degree + 1) ** 2 or K >= self._features.shape[1]:
            raise ValueError("begin_dense_dc: K = %d coefficients cannot carry SH degree %d of %d-coefficient rows densely" % (
                K, self.active_sh_degree, self._features.shape[1]))
        if self._features_dc0 is not None and self._features_dc0.shape[1] != K:
            self.end_dense_dc()
        if self._features_dc0 is None:
            self._set_dense_dc(torch.nn.Parameter(self._features.detach()[:, :K].clone().contiguous(), requires_grad=self._features.requires_grad))
        return self._features_dc0

    def _set_dense_dc(self, leaf):
        """The dense leaf is held as a plain attribute (nn.Module.__setattr__ would register a Parameter under a key a fresh model
        does not have: strict load_state_dict would fail on it and a non-strict one would restore a stale coefficient 0)."""
        object.__setattr__(self, "_features_dc0", leaf)

    def end_dense_dc(self):
        """Fold the dense coefficient-0 leaf back into the [N,16,3] rows (in place: shared storage and views stay valid)."""
        if self._features_dc0 is not None:
            with torch.no_grad():
                self._features[:, :self._features_dc0.shape[1]].copy_(self._features_dc0)
            self._set_dense_dc(None)
        return self._features

    def state_dict(self, *args, **kwargs):
        """nn.Module.state_dict with the rows CURRENT: while the dense leaf is trained, coefficient 0 of `_features` is stale, so the
        serialised rows are written from get_features (the live tensors are untouched)."""
        sd = super().state_dict(*args, **kwargs)
        if self._features_dc0 is not None:
            key = (kwargs.get("prefix", args[1] if len(args) > 1 else "") or "") + "_features"
            if key in sd:
                rows = sd[key].detach().clone()
                rows[:, :self._features_dc0.shape[1]].copy_(self._features_dc0.detach())
                sd[key] = rows
        return sd

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def _features_dc(self):
        return self._features[:, :1] if self._features_dc0 is None else self._features_dc0[:, :1]

    @property
    def _features_rest(self):
        return self.get_features[:, 1:] if self._features_dc0 is not None and self._features_dc0.shape[1] > 1 else self._features[:, 1:]

    @property
    def get_features(self):
        if self._features_dc0 is not None:       # (the python SH route and external readers; the HIP route takes renderer.sh_operand)
            return torch.cat((self._features_dc0, self._features[:, self._features_dc0.shape[1]:].detach()), dim=1)
        return self._features

    def activated(self, mr_weight=None, joint=None):
        """(get_xyz, get_scaling, get_rotation, get_opacity[, mesh_restrict_loss]) from one fused kernel
        (gm_mesh_activate_fwd / _bwd); joint: see model_ops.mesh_activate."""
        from .model_ops import mesh_activate
        return mesh_activate(self._bc, self._distance, self._scaling, self._rotation, self._opacity, self.vertex1, self.vertex2,
                             self.vertex3, self.normal, self.r, float(self.alpha_distance), mr_weight, joint)

    @property
    def get_proj_xyz(self):
        bc = torch.softmax(self._bc, dim=1)
        return bc[:, 0:1] * self.vertex1 + bc[:, 1:2] * self.vertex2 + bc[:, 2:3] * self.vertex3

    @property
    def get_xyz(self):
        return self.get_proj_xyz + self.alpha_distance * self.r * (torch.sigmoid(self._distance) - 0.5) * self.normal

    def get_covariance(self, scaling_modifier=1):
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                         2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        return strip_symmetric(L @ L.transpose(1, 2))
