"""GPU: the fixed-topology training iteration (gaussianmesh_amd/train.py) end to end: rasterizer forward+backward,
L1+SSIM loss kernels, mesh-restrict loss, Adam - the loss goes down and the image approaches the target."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(N, seed, perturb):
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
    shs = cl["shs"].copy()
    opac = np.full((N, 1), 1.0, np.float32)
    if perturb:                                         # the student starts from grey, half-transparent Gaussians
        shs[:] = 0.0
        opac[:] = -1.0
    return MeshBoundGaussians(T(np.zeros((N, 3))), T(np.zeros((N, 1))), T(shs[:, :1]), T(shs[:, 1:]), T(np.log(cl["scales"] * 6)),
                              T(cl["rots"]), T(opac), T(v1), T(v2), T(v3), T(n), T(r)).cuda()


def test_training_iterations_reduce_the_loss():
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, render
    from gaussianmesh_amd.train import Trainer
    from types import SimpleNamespace
    N = 4000
    cams = [Camera(scenes.orbit_camera(k, 4, 192, 128, radius=7.0), "cuda") for k in range(4)]
    bg = torch.zeros(3, device="cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    teacher = _model(N, 0, perturb=False)
    with torch.no_grad():
        targets = [render(c, teacher, pipe, bg)["render"].clone() for c in cams]
    student = _model(N, 0, perturb=True)
    tr = Trainer(student, alpha_mrloss=6.0, feature_lr=0.02, opacity_lr=0.1)
    losses = []
    for it in range(60):
        loss, pkg = tr.step(cams[it % 4], targets[it % 4], bg)
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    first, last = np.mean(losses[:4]), np.mean(losses[-4:])
    assert last < 0.6 * first, (first, last)
    assert pkg["render"].shape == (3, 128, 192) and pkg["radii"].shape == (N,)
    # every parameter group moved
    fresh = _model(N, 0, perturb=True)
    for name in ("_bc", "_distance", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert not torch.equal(getattr(student, name), getattr(fresh, name)), name
    assert tr.optimizer.param_groups[0]["lr"] < tr.opt.position_lr_init          # schedule is applied


def test_trainer_sync_free_densify_stats_and_background_cloud():
    """Trainer(sync_free=True, densify_stats=True, bg_gaussian=...): the same parameter trajectory as the exact-count
    trainer (bit-identical images and losses: the instance count only sizes a buffer), the densification statistics of
    train_mesh_gaussian.py:119-126 against their torch statement, and an overflowing iteration is redone, not lost."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.renderer import Camera, MeshBoundGaussians
    from gaussianmesh_amd.train import FrozenGaussians, Trainer
    verts, faces = scenes.torus_mesh(24, 16)
    N = 4000

    def build():
        cl = scenes.bind_cloud_to_mesh(N, verts, faces, seed=2)
        tri = faces[cl["fid"]]
        v1, v2, v3 = (T(verts[tri[:, k]].astype(np.float32)) for k in range(3))
        nrm = torch.nn.functional.normalize(torch.cross(v2 - v1, v3 - v1, dim=1), dim=1)
        rad = (((v2 - v1).norm(dim=1) + (v3 - v2).norm(dim=1) + (v1 - v3).norm(dim=1)) / 3)[:, None]
        return MeshBoundGaussians(torch.log(T(cl["weights"]).clamp_min(1e-6)), torch.zeros((N, 1), device="cuda"), T(cl["shs"][:, :1]),
                                  T(cl["shs"][:, 1:]), torch.log(T(cl["scales"] * 4)), T(cl["rots"]), torch.logit(T(cl["opac"]).reshape(-1, 1)),
                                  v1, v2, v3, nrm, rad).cuda()
    b = scenes.make_cloud(500, seed=9, scale_lo=0.05, scale_hi=0.3)
    nb = np.linalg.norm(b["means"], axis=1, keepdims=True) + 1e-6
    bg = FrozenGaussians(T(b["means"] / nb * (4 + nb)), T(b["scales"]), torch.nn.functional.normalize(T(b["rots"])), T(b["opac"]).reshape(-1, 1), T(b["shs"]))
    cams = [Camera(scenes.orbit_camera(k, 5, 160, 96, radius=6.5), "cuda") for k in range(5)]
    gt = torch.rand((3, 96, 160), device="cuda")
    zero = torch.zeros(3, device="cuda")
    ta = Trainer(build(), densify_stats=True, sync_free=False, bg_gaussian=bg)
    tb = Trainer(build(), densify_stats=True, sync_free=True, bg_gaussian=bg)
    max_r = torch.zeros(N, device="cuda"); acc = torch.zeros((N, 1), device="cuda"); den = torch.zeros((N, 1), device="cuda")
    for i in range(6):
        if i == 4:
            Rz._sync_free["capacity"][torch.device("cuda", 0)] = 64          # far too small: iteration 4 of tb must be redone
        la, pa = ta.step(cams[i % 5], gt, zero)
        vs_grad = ta.g.screenspace_points.grad                                # cleared at the start of the next step only
        lb, pb = tb.step(cams[i % 5], gt, zero)
        if i == 0:                      # same parameters: bit-identical image (the instance count only sizes a buffer) ...
            assert torch.equal(pa["render"], pb["render"]) and torch.equal(la, lb)
        else:                           # ... afterwards the two runs differ by the float-atomic summation order of their backward passes
            assert (pa["render"] - pb["render"]).abs().max() <= 2e-2 and abs(float(la) - float(lb)) <= 1e-3 * abs(float(la)), i
        vis = pa["radii"][:N] > 0
        max_r[vis] = torch.maximum(max_r[vis], pa["radii"][:N][vis].float())
    assert tb.redone == 1 and ta.redone == 0
    for p, q in zip(ta.g.parameters(), tb.g.parameters()):
        assert (p - q).abs().max() <= 5e-3 * max(float(p.abs().max()), 1.0)
    assert torch.equal(ta.max_radii2D, max_r)
    assert (ta.max_radii2D - tb.max_radii2D).abs().max() <= 1 and (ta.denom - tb.denom).abs().max() <= 1      # radii may flip by one
    assert ta.denom.max() <= 6 and ta.denom.sum() > 0
    assert torch.isfinite(ta.bc_gradient_accum).all() and (ta.bc_gradient_accum[ta.denom > 0] >= 0).all()
    assert pa["radii"].shape[0] == N + 500 and pa["scale"].shape[0] == N


def test_densify_stats_kernel_matches_the_reference_statements():
    from gaussianmesh_amd.model_ops import densify_stats
    g = torch.Generator(device="cuda").manual_seed(0)
    N = 10007
    radii = torch.randint(-2, 40, (N,), device="cuda", generator=g, dtype=torch.int32)
    grad = torch.randn((N, 3), device="cuda", generator=g)
    mr = torch.rand(N, device="cuda", generator=g) * 30; acc = torch.rand((N, 1), device="cuda", generator=g); den = torch.zeros((N, 1), device="cuda")
    mr0, acc0, den0 = mr.clone(), acc.clone(), den.clone()
    densify_stats(radii, grad, mr, acc, den)
    vis = radii > 0
    mr0[vis] = torch.maximum(mr0[vis], radii[vis].float())                    # train_mesh_gaussian.py:123
    acc0[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)               # mesh_based_gaussian_model.py:588
    den0[vis] += 1
    assert torch.equal(mr, mr0) and torch.equal(den, den0) and (acc - acc0).abs().max() <= 1e-6


def test_shared_feature_storage_equals_concatenation():
    """render(..., bg_gaussian=...) with the SH parameter living in one buffer with the background's rows
    (renderer.share_feature_storage: no per-iteration torch.cat) gives the image and the gradients of the concatenating path,
    and an in-place optimizer step on the parameter is seen by the next render."""
    from types import SimpleNamespace
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, render, share_feature_storage
    from gaussianmesh_amd.train import FrozenGaussians
    b = scenes.make_cloud(700, seed=4, scale_lo=0.05, scale_hi=0.3)
    bg = FrozenGaussians(T(b["means"] * 1.5 + np.array([0, 0, 1.0], np.float32)), T(b["scales"]), torch.nn.functional.normalize(T(b["rots"])),
                         T(b["opac"]).reshape(-1, 1), T(b["shs"]))
    cam = Camera(scenes.orbit_camera(2, 7, 160, 96, radius=7.0), "cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    zero = torch.zeros(3, device="cuda")
    w = torch.rand((3, 96, 160), device="cuda")
    ma, mb = _model(1500, 0, perturb=False), _model(1500, 0, perturb=False)
    share_feature_storage(mb, bg)
    assert mb._features.is_leaf and mb._features.requires_grad and torch.equal(ma._features, mb._features)
    outs = []
    for m in (ma, mb):
        img = render(cam, m, pipe, zero, bg_gaussian=bg)["render"]
        (img * w).sum().backward()
        outs.append((img.detach(), m._features.grad.clone(), m._opacity.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert (outs[0][1] - outs[1][1]).abs().max() <= 1e-5 * outs[0][1].abs().max() and (outs[0][2] - outs[1][2]).abs().max() <= 1e-5 * outs[0][2].abs().max()
    with torch.no_grad():                                   # what an optimizer does
        for m in (ma, mb):
            m._features.add_(0.05)
    ia = render(cam, ma, pipe, zero, bg_gaussian=bg)["render"]
    ib = render(cam, mb, pipe, zero, bg_gaussian=bg)["render"]
    assert torch.equal(ia, ib) and not torch.equal(ia.detach(), outs[0][0])
