"""GPU: the training loop across a topology change (scene/mesh_based_gaussian_model.py:411-563, 596-647 on the
HIP ops): the number of Gaussians changes between two iterations and rasterizer, fused activations, FusedAdam, densification
statistics, shared SH storage and the sync-free forward all follow.  The row semantics of the edits themselves are pinned
against reference-executed fixtures on the CPU (tests/test_topology.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bg(n=600):
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.train import FrozenGaussians
    b = scenes.make_cloud(n, seed=9, scale_lo=0.05, scale_hi=0.3)
    nb = np.linalg.norm(b["means"], axis=1, keepdims=True) + 1e-6
    return FrozenGaussians(T(b["means"] / nb * (4 + nb)), T(b["scales"]), torch.nn.functional.normalize(T(b["rots"])), T(b["opac"]).reshape(-1, 1),
                           T(b["shs"]))


def _snapshot(tr):
    return {g["name"]: (g["params"][0].detach().clone(), g["m"][0].clone(), g["values"][0].clone()) for g in tr.optimizer.param_groups}


def test_split_and_prune_between_iterations():
    """20 iterations, then 10 % of the faces are split 1 -> 4 (densify_and_split, :508-563) and 5 % of the rows pruned
    (prune_points, :440-463), then 20 more iterations - with a frozen background cloud (shared SH storage re-bound), densification
    statistics and the sync-free forward.  Surviving rows keep parameter AND both Adam moments bit for bit, appended rows start
    from zero moments, the loss keeps going down, every one of the 40 iterations took its optimizer step."""
    from test_gpu_train import _model
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, render
    from gaussianmesh_amd.train import Trainer
    from types import SimpleNamespace
    N = 4000
    cams = [Camera(scenes.orbit_camera(k, 4, 192, 128, radius=7.0), "cuda") for k in range(4)]
    zero = torch.zeros(3, device="cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    bg = _bg()
    teacher = _model(N, 0, perturb=False)
    with torch.no_grad():
        targets = [render(c, teacher, pipe, zero, bg_gaussian=bg)["render"].clone() for c in cams]
    student = _model(N, 0, perturb=True)
    tr = Trainer(student, alpha_mrloss=6.0, feature_lr=0.02, opacity_lr=0.1, densify_stats=True, sync_free=True, bg_gaussian=bg)
    losses = []
    for it in range(20):
        loss, pkg = tr.step(cams[it % 4], targets[it % 4], zero)
        losses.append(loss)
    assert pkg["radii"].shape[0] == N + 600
    assert float(tr.denom.max()) > 0
    g = torch.Generator(device="cuda").manual_seed(3)
    # ---- split 10 % of the faces into four
    before = _snapshot(tr)
    sel = torch.rand(N, device="cuda", generator=g) < 0.10
    ns = int(sel.sum())
    area = lambda a, b, c: torch.linalg.cross(b - a, c - a).norm(dim=1) / 2
    parent_area = area(tr.g.vertex1[sel], tr.g.vertex2[sel], tr.g.vertex3[sel])
    n1 = tr.densify_and_split(sel, N=4)
    assert n1 == N - ns + 4 * ns and tr.g.get_number == n1
    keep = (~sel).nonzero().reshape(-1)
    for name, (p0, m0, v0) in before.items():
        grp = [q for q in tr.optimizer.param_groups if q["name"] == name][0]
        p1, m1, v1 = grp["params"][0], grp["m"][0], grp["values"][0]
        assert p1.shape[0] == n1 and p1.is_leaf and p1.requires_grad
        assert torch.equal(p1[:N - ns], p0[keep]) and torch.equal(m1[:N - ns], m0[keep]) and torch.equal(v1[:N - ns], v0[keep]), name
        assert not m1[N - ns:].any() and not v1[N - ns:].any(), name                           # appended rows: zero moments
        assert m0[keep].abs().sum() > 0                                                         # (the moments were not trivially zero)
    assert torch.equal(tr.g._bc[N - ns:], torch.full((4 * ns, 3), 1.0 / 3.0, device="cuda"))
    assert tr.g._features.data_ptr() == tr.g._features_with_bg[0].data_ptr()                 # SH rows back in the shared storage
    assert tr.max_radii2D.shape == (n1,) and not tr.max_radii2D.any() and not tr.denom.any()   # :503-505
    # the four children tile their parent: same area in total
    child_area = area(tr.g.vertex1[N - ns:], tr.g.vertex2[N - ns:], tr.g.vertex3[N - ns:]).reshape(ns, 4).sum(1)
    assert torch.allclose(child_area, parent_area, rtol=1e-4)
    # ---- one iteration on the new topology, then prune 5 %
    loss, pkg = tr.step(cams[0], targets[0], zero)
    losses.append(loss)
    assert pkg["radii"].shape[0] == n1 + 600 and pkg["scale"].shape[0] == n1 and tr.g.screenspace_points.shape[0] == n1
    before = _snapshot(tr)
    stats_before = (tr.max_radii2D.clone(), tr.bc_gradient_accum.clone(), tr.denom.clone())
    mask = torch.rand(n1, device="cuda", generator=g) < 0.05
    n2 = tr.prune_points(mask)
    assert n2 == n1 - int(mask.sum())
    keep = (~mask).nonzero().reshape(-1)
    for name, (p0, m0, v0) in before.items():
        grp = [q for q in tr.optimizer.param_groups if q["name"] == name][0]
        assert torch.equal(grp["params"][0], p0[keep]) and torch.equal(grp["m"][0], m0[keep]) and torch.equal(grp["values"][0], v0[keep]), name
    for now, old in zip((tr.max_radii2D, tr.bc_gradient_accum, tr.denom), stats_before):       # :452-455: statistics of survivors are kept
        assert torch.equal(now, old[keep])
    for it in range(19):
        loss, pkg = tr.step(cams[(it + 1) % 4], targets[(it + 1) % 4], zero)
        losses.append(loss)
    losses = [float(l) for l in losses]
    assert len(losses) == 40 and all(np.isfinite(losses))
    assert tr.optimizer.n_step == 40 and tr.iteration == 40 and tr.resizes == 2                 # no iteration lost (redone ones are repeated)
    assert np.mean(losses[16:20]) < 0.8 * np.mean(losses[:4])                                   # before the edit: going down
    assert np.mean(losses[-4:]) < np.mean(losses[21:25])                                        # after the edit: still going down
    assert np.mean(losses[-4:]) < np.mean(losses[:4])
    assert pkg["radii"].shape[0] == n2 + 600


def test_fused_adam_through_a_resize_matches_a_torch_adam_twin():
    """FusedAdam (jittor.nn.Adam's rule) against torch.optim.Adam on identical synthetic gradients, with the optimizer surgery of
    scene/mesh_based_gaussian_model.py:425-438 / :465-483 done by hand on the twin (state rows follow the parameters, new rows
    get zero moments AND keep the group's step count, as in Jittor where the step count is global): parameters and both
    moments agree to float rounding before and after."""
    from gaussianmesh_amd.model_ops import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(0)
    n, lr, eps = 5000, 0.01, 1e-15
    p = torch.nn.Parameter(torch.randn((n, 3), device="cuda", generator=g))
    q = torch.nn.Parameter(p.detach().clone())
    fa = FusedAdam([{"params": [p], "lr": lr, "name": "bc"}], eps=eps)
    ta = torch.optim.Adam([q], lr=lr, eps=eps)

    def both(k):
        for _ in range(k):
            gr = torch.randn(fa.param_groups[0]["params"][0].shape, device="cuda", generator=g)
            fa.param_groups[0]["params"][0].grad = gr.clone()
            ta.param_groups[0]["params"][0].grad = gr.clone()
            fa.step(); ta.step()

    def check():
        a = fa.param_groups[0]; b = ta.param_groups[0]["params"][0]; st = ta.state[b]
        assert (a["params"][0] - b).abs().max() <= 2e-6 * b.abs().max()
        assert (a["m"][0] - st["exp_avg"]).abs().max() <= 1e-6 * st["exp_avg"].abs().max()
        assert (a["values"][0] - st["exp_avg_sq"]).abs().max() <= 1e-6 * st["exp_avg_sq"].abs().max()
    both(7); check()
    keep = torch.rand(n, device="cuda", generator=g) < 0.8
    ext = torch.randn((300, 3), device="cuda", generator=g)
    fa.resize(keep=keep, new_rows={"bc": ext})
    old = ta.param_groups[0]["params"][0]; st = ta.state.pop(old)
    new = torch.nn.Parameter(torch.cat((old.detach()[keep], ext)))
    ta.param_groups[0]["params"][0] = new
    ta.state[new] = {"step": st["step"], "exp_avg": torch.cat((st["exp_avg"][keep], torch.zeros_like(ext))),
                     "exp_avg_sq": torch.cat((st["exp_avg_sq"][keep], torch.zeros_like(ext)))}
    check()
    both(6); check()
    assert fa.param_groups[0]["params"][0].shape[0] == int(keep.sum()) + 300
