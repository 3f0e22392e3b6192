"""GPU: the background-cloud render routes of SURVEY.md 8 f1 against the CPU oracle.

  render(cam, pc, pipe, bg, bg_gaussian=...)       gaussian_renderer/__init__.py:100-121  (frozen free cloud behind the mesh-bound model)
  bg_render(cam, pc, pipe, bg, mesh_gaussians=...) gaussian_renderer/__init__.py:146-260  (free cloud trained, mesh-bound model frozen)

What is compared: the image through the strict flip-accounted forward gate, radii[N + Nb] bit for bit, the "scale" entry, and the
gradients that reach the MODEL PARAMETERS.  Reference gradients: oracle.backward_full on the concatenated scene gives dL/d(means,
scales, rotations | cov3D, opacity, shs | colours) per row; the trainable rows are chained to the parameters through a float64 torch
statement of the activations (scene/mesh_based_gaussian_model.py:122-174) differentiated by autograd - nothing of the product's
own backward is on the reference side.  All four (compute_cov3D_python, convert_SHs_python) routes, and the shared SH storage the
Trainer uses."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import assert_forward_gate, assert_grads_elementwise

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_RTOL = 1e-3
N, NB, W, H = 2500, 700, 176, 112


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _grad_gate(got, ref, what):
    assert np.isfinite(np.asarray(got)).all(), what
    assert _rel(got, ref) <= GRAD_RTOL, (what, _rel(got, ref))
    assert_grads_elementwise(got, ref, what)


def _mesh_model(seed=3):
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import MeshBoundGaussians
    verts, faces = scenes.torus_mesh(24, 16)
    rng = np.random.default_rng(seed)
    cl = scenes.bind_cloud_to_mesh(N, verts, faces, seed=2)
    tri = faces[cl["fid"]]
    v1, v2, v3 = (verts[tri[:, k]].astype(np.float32) for k in range(3))
    n = np.cross(v2 - v1, v3 - v1); n /= np.linalg.norm(n, axis=1, keepdims=True)
    r = ((np.linalg.norm(v2 - v1, axis=1) + np.linalg.norm(v3 - v2, axis=1) + np.linalg.norm(v1 - v3, axis=1)) / 3)[:, None]
    return MeshBoundGaussians(T(rng.normal(size=(N, 3))), T(rng.normal(0, 0.3, size=(N, 1))), T(cl["shs"][:, :1]), T(cl["shs"][:, 1:]),
                              T(np.log(cl["scales"] * 6)), T(cl["rots"] * rng.uniform(0.5, 2.0, size=(N, 1))), T(rng.normal(size=(N, 1))),
                              T(v1), T(v2), T(v3), T(n), T(r)).cuda()


def _shell(seed=9):
    """free cloud on a shell around the torus, part of it between the camera and the object"""
    from gaussianmesh_amd import scenes
    b = scenes.make_cloud(NB, seed=seed, scale_lo=0.05, scale_hi=0.3)
    nb = np.linalg.norm(b["means"], axis=1, keepdims=True) + 1e-6
    b["means"] = (b["means"] / nb * (3.0 + nb)).astype(np.float32)
    return b


class _FreeGaussians(torch.nn.Module):
    """the plain (not mesh-bound) model of scene/gaussian_model.py:24-97 as far as bg_render reads it: parameters and activations"""

    def __init__(self, b):
        super().__init__()
        from gpu_utils import T
        P = torch.nn.Parameter
        self._xyz = P(T(b["means"]))
        self._scaling = P(torch.log(T(b["scales"])))
        self._rotation = P(T(b["rots"]) * 1.7)
        self._opacity = P(torch.logit(T(b["opac"]).reshape(-1, 1)))
        self._features = P(T(b["shs"]))
        self.active_sh_degree = self.max_sh_degree = 3
        self.screenspace_points = torch.zeros_like(self._xyz, requires_grad=True)

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: s._features)

    def get_covariance(self, scaling_modifier=1):
        return _cov6(self.get_scaling * scaling_modifier, self.get_rotation)


def _cov6(s, q):
    """strip_symmetric(R diag(s)^2 R^T) (utils/general_utils.py:64-131) for unit quaternions q = (r, x, y, z); any float dtype"""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                     2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * s[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def _d(t):
    return t.detach().double().clone().requires_grad_(True)


def _mesh_f64(pc):
    """float64 leaves of the mesh-bound parameters and their activations (mesh_based_gaussian_model.py:122-152, 172-174)"""
    L = dict(bc=_d(pc._bc), dist=_d(pc._distance), scaling=_d(pc._scaling), rot=_d(pc._rotation), opac=_d(pc._opacity), feat=_d(pc._features))
    w = torch.softmax(L["bc"], dim=1)
    v1, v2, v3, n, r = (t.double() for t in (pc.vertex1, pc.vertex2, pc.vertex3, pc.normal, pc.r))
    xyz = w[:, 0:1] * v1 + w[:, 1:2] * v2 + w[:, 2:3] * v3 + pc.alpha_distance * r * (torch.sigmoid(L["dist"]) - 0.5) * n
    act = dict(means=xyz, scales=torch.exp(L["scaling"]), rots=torch.nn.functional.normalize(L["rot"]), opac=torch.sigmoid(L["opac"]), shs=L["feat"])
    return L, act


def _free_f64(pc):
    L = dict(xyz=_d(pc._xyz), scaling=_d(pc._scaling), rot=_d(pc._rotation), opac=_d(pc._opacity), feat=_d(pc._features))
    act = dict(means=L["xyz"], scales=torch.exp(L["scaling"]), rots=torch.nn.functional.normalize(L["rot"]), opac=torch.sigmoid(L["opac"]), shs=L["feat"])
    return L, act


def _python_colors_f64(act, campos, deg):
    from gaussianmesh_amd.renderer import eval_sh_torch            # the polynomial pinned by tests/golden/sh_eval.npz
    d = act["means"] - campos.double().reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh_torch(deg, act["shs"], d) + 0.5, 0.0)


def _frozen_arrays(oracle, other, campos):
    """numpy float32 rows of the frozen cloud as the render glue hands them to the op; `other` = dict of numpy arrays"""
    d = other["means"] - campos.reshape(1, 3)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    col = np.maximum(oracle.sh_to_rgb(3, other["shs"], d) + 0.5, 0.0).astype(np.float32)
    q = torch.as_tensor(other["rots"]).double(); q = q / q.norm(dim=1, keepdim=True)
    cov = _cov6(torch.as_tensor(other["scales"]).double(), q).numpy().astype(np.float32)
    return dict(other, colors_precomp=col, cov3D_precomp=cov)


def _oracle_route(oracle, act, frozen, cam_d, bg, dpix, cov_py, sh_py, deg, op_in=None):
    """Oracle forward + backward on [trainable rows; frozen rows]; returns (fw, reference parameter-side gradients by chaining the
    trainable rows' gradients through the float64 graph `act` hangs from, oracle gradient dict).
    op_in: the float32 rows the op was actually handed for the trainable part ({"means","opac","scales","rots","cov","col"} as
    present) - the oracle then sees the SAME inputs as the HIP path (bit-identical geometry is a claim about equal inputs; the
    activations themselves are checked against float64 next to it, tests/test_gpu_model_ops.py and below); the gradient chain
    always runs through the float64 activations."""
    n = act["means"].shape[0]
    op_in = op_in or {}

    def f32(t, key=None):
        if key is not None and op_in.get(key) is not None:
            got = op_in[key].detach().cpu().numpy().astype(np.float32).reshape(tuple(t.shape))
            ref = t.detach().cpu().numpy()
            assert np.abs(got - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0), ("activation", key, np.abs(got - ref).max())
            return got
        return t.detach().cpu().numpy().astype(np.float32)
    campos = torch.as_tensor(cam_d["campos"])
    cat = lambda a, b: np.concatenate([a, b], axis=0)
    sc = dict(means=cat(f32(act["means"], "means"), frozen["means"]), opac=cat(f32(act["opac"], "opac"), frozen["opac"].reshape(-1, 1)))
    col = cov = None
    if sh_py:
        col = _python_colors_f64(act, campos.to(act["means"].device), deg)
        sc["colors_precomp"] = cat(f32(col, "col"), frozen["colors_precomp"])
    else:
        sc["shs"] = cat(f32(act["shs"]), frozen["shs"])
    if cov_py:
        cov = _cov6(act["scales"], act["rots"])
        sc["cov3D_precomp"] = cat(f32(cov, "cov"), frozen["cov3D_precomp"])
    else:
        sc["scales"] = cat(f32(act["scales"], "scales"), frozen["scales"]); sc["rots"] = cat(f32(act["rots"], "rots"), frozen["rots"])
    fw = oracle.forward_full(sc, cam_d, bg, D=deg, use_precomp_cov=cov_py, use_precomp_color=sh_py)
    bw = oracle.backward_full(sc, cam_d, bg, fw, dpix, D=deg, use_precomp_cov=cov_py, use_precomp_color=sh_py)
    dev = act["means"].device
    t = lambda a: torch.as_tensor(np.asarray(a[:n], np.float64), device=dev)
    obj = (act["means"] * t(bw["dmean3D"])).sum() + (act["opac"] * t(bw["dopacity"]).reshape(-1, 1)).sum()
    obj = obj + ((col * t(bw["dcolor"])).sum() if sh_py else (act["shs"] * t(bw["dsh"])).sum())
    obj = obj + ((cov * t(bw["dcov3D"])).sum() if cov_py else ((act["scales"] * t(bw["dscale"])).sum() + (act["rots"] * t(bw["drot"])).sum()))
    return fw, obj, bw, sc


ROUTES = [(False, False), (True, False), (False, True), (True, True)]


@pytest.mark.parametrize("cov_py,sh_py", ROUTES)
@pytest.mark.parametrize("shared", [False, True])
def test_render_with_background_cloud_vs_oracle(oracle, cov_py, sh_py, shared):
    """render(..., bg_gaussian=FrozenGaussians): image, radii, scale and the gradients reaching _bc / _distance / _scaling /
    _rotation / _opacity / _features and the view-space points, on all four pipe routes; `shared`: the SH parameter lives in one
    buffer with the background's rows (renderer.share_feature_storage, what Trainer sets up)."""
    if shared and sh_py:
        pytest.skip("shared SH storage only concerns the SH-in-the-op route")
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, render, share_feature_storage
    from gaussianmesh_amd.train import FrozenGaussians
    pc = _mesh_model()
    pc.active_sh_degree = 2                                           # the op evaluates BOTH clouds at the model's active degree (:59, :117)
    b = _shell()
    frozen_t = FrozenGaussians(T(b["means"]), T(b["scales"]), torch.nn.functional.normalize(T(b["rots"])), T(b["opac"]).reshape(-1, 1), T(b["shs"]))
    if shared:
        share_feature_storage(pc, frozen_t)
    cam_d = scenes.orbit_camera(1, 6, W, H, radius=6.5)
    cam = Camera(cam_d, "cuda")
    bg = np.array([0.7, 0.1, 0.4], np.float32)                        # train_mesh_gaussian.py:85: a random background when a bg cloud exists
    pipe = SimpleNamespace(convert_SHs_python=sh_py, compute_cov3D_python=cov_py, debug=False)
    rng = np.random.default_rng(5)
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    out = render(cam, pc, pipe, T(bg), bg_gaussian=frozen_t)
    (out["render"] * T(dpix)).sum().backward()
    torch.cuda.synchronize()
    # ---- reference
    L, act = _mesh_f64(pc)
    frozen = _frozen_arrays(oracle, dict(b, rots=frozen_t.get_rotation.cpu().numpy()), cam_d["campos"])
    from gaussianmesh_amd import renderer as Rn
    op_in = {}
    if cov_py:
        op_in.update(means=pc.get_xyz, opac=pc.get_opacity, cov=pc.get_covariance(1.0))
    else:
        xyz, sca, rot, opa = pc.activated()[:4]
        op_in.update(means=xyz, opac=opa, scales=sca, rots=rot)
    if sh_py:
        op_in["col"] = Rn._python_sh_colors(pc, cam, op_in["means"], pc.get_features)
    fw, obj, bw, sc = _oracle_route(oracle, act, frozen, cam_d, bg, dpix, cov_py, sh_py, pc.active_sh_degree, op_in)
    obj.backward()
    what = "render+bg cov_py=%d sh_py=%d shared=%d" % (cov_py, sh_py, shared)
    assert out["radii"].shape == (N + NB,) and out["viewspace_points"].shape == (N + NB, 3)
    radii = out["radii"].cpu().numpy()
    assert (radii[:N] > 0).sum() > N // 4 and (radii[N:] > 0).sum() > NB // 8        # both clouds are in the picture
    assert np.array_equal(radii, fw["geo"]["radii"]), what            # same float32 inputs: bit-identical geometry
    if not cov_py:
        assert out["scale"].shape == (N, 3) and float((out["scale"].detach().double() - act["scales"].detach()).abs().max()) <= 2e-6 * float(act["scales"].detach().abs().max())
    else:
        assert out["scale"] is None                                                   # :143 returns the (unset) scales
    assert_forward_gate(fw, out["render"].detach().cpu().numpy(), W, H, FWD_TOL, what)
    got = dict(bc=pc._bc.grad, dist=pc._distance.grad, scaling=pc._scaling.grad, rot=pc._rotation.grad, opac=pc._opacity.grad, feat=pc._features.grad)
    for k, g in got.items():
        assert g is not None and tuple(g.shape) == tuple(L[k].shape), (what, k)
        _grad_gate(g.cpu().numpy(), L[k].grad.cpu().numpy(), what + " d/d" + k)
    if shared:                                                        # the frozen rows of the shared buffer got no gradient of their own
        assert pc._features.grad.shape[0] == N
    vs = pc.screenspace_points.grad                                   # dL/dmeans2D of the trainable rows (densification statistic input)
    assert vs is not None and vs.shape == (N, 3)
    _grad_gate(vs.cpu().numpy()[:, :2], bw["dmean2D"][:N, :2], what + " d/dmeans2D")
    # the frozen cloud received nothing
    for t in (frozen_t.get_xyz, frozen_t.get_scaling, frozen_t.get_rotation, frozen_t.get_opacity, frozen_t.get_features):
        assert t.grad is None


@pytest.mark.parametrize("cov_py,sh_py", [(False, False)])
def test_bg_render_with_frozen_mesh_gaussians_vs_oracle(oracle, cov_py, sh_py):
    """bg_render(..., mesh_gaussians=pc_mesh) (:146-260): the free cloud is trained, the mesh-bound model rides along detached
    (.stop_grad() at :229-233; scales / rotations / SH concatenated, so this route exists for the in-op covariance and colour only)."""
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, bg_render
    mesh = _mesh_model()
    b = _shell(seed=11)
    pc = _FreeGaussians(b).cuda()
    cam_d = scenes.orbit_camera(4, 6, W, H, radius=6.5)
    cam = Camera(cam_d, "cuda")
    bg = np.array([0.2, 0.9, 0.5], np.float32)
    pipe = SimpleNamespace(convert_SHs_python=sh_py, compute_cov3D_python=cov_py, debug=False)
    dpix = np.random.default_rng(6).normal(size=(3, H, W)).astype(np.float32)
    out = bg_render(cam, pc, pipe, T(bg), mesh_gaussians=mesh)
    (out["render"] * T(dpix)).sum().backward()
    torch.cuda.synchronize()
    L, act = _free_f64(pc)
    with torch.no_grad():
        frozen = dict(means=mesh.get_xyz.cpu().numpy(), scales=mesh.get_scaling.cpu().numpy(), rots=mesh.get_rotation.cpu().numpy(),
                      opac=mesh.get_opacity.cpu().numpy(), shs=mesh.get_features.cpu().numpy())
    frozen = _frozen_arrays(oracle, frozen, cam_d["campos"])
    fw, obj, bw, sc = _oracle_route(oracle, act, frozen, cam_d, bg, dpix, cov_py, sh_py, 3)
    obj.backward()
    what = "bg_render+mesh"
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
    assert out["radii"].shape == (NB + N,) and np.array_equal(out["radii"].cpu().numpy(), fw["geo"]["radii"])
    assert torch.equal(out["visibility_filter"], out["radii"] > 0)
    assert_forward_gate(fw, out["render"].detach().cpu().numpy(), W, H, FWD_TOL, what)
    got = dict(xyz=pc._xyz.grad, scaling=pc._scaling.grad, rot=pc._rotation.grad, opac=pc._opacity.grad, feat=pc._features.grad)
    for k, g in got.items():
        assert g is not None, k
        _grad_gate(g.cpu().numpy(), L[k].grad.cpu().numpy(), what + " d/d" + k)
    for p in mesh.parameters():                                       # .stop_grad(): nothing reaches the mesh-bound model
        assert p.grad is None
    _grad_gate(pc.screenspace_points.grad.cpu().numpy()[:, :2], bw["dmean2D"][:NB, :2], what + " d/dmeans2D")
