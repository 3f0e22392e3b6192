"""GPU parity of the fused training-loop operators (csrc/gm_train.hip) against the plain torch composition of the
reference's formulas (scene/mesh_based_gaussian_model.py:122-152, 172-174) and against torch.optim.Adam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(N, seed):
    rng = np.random.default_rng(seed)
    t = lambda a, g=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=g)
    n = rng.standard_normal((N, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    return dict(bc=t(rng.standard_normal((N, 3)) * 2, True), dist=t(rng.standard_normal((N, 1)), True),
                scaling=t(rng.standard_normal((N, 3)) - 2, True), rot=t(rng.standard_normal((N, 4)), True),
                opac=t(rng.standard_normal((N, 1)) * 2, True), v1=t(rng.standard_normal((N, 3))), v2=t(rng.standard_normal((N, 3))),
                v3=t(rng.standard_normal((N, 3))), n=t(n), r=t(rng.random((N, 1)) + 0.1))


def _torch_ref(d):
    w = torch.softmax(d["bc"].double(), dim=1)
    xyz = w[:, 0:1] * d["v1"].double() + w[:, 1:2] * d["v2"].double() + w[:, 2:3] * d["v3"].double()
    xyz = xyz + 4 * d["r"].double() * (torch.sigmoid(d["dist"].double()) - 0.5) * d["n"].double()
    return xyz, torch.exp(d["scaling"].double()), torch.nn.functional.normalize(d["rot"].double()), torch.sigmoid(d["opac"].double())


@pytest.mark.parametrize("N", [1, 257, 10000])
def test_mesh_activate_forward_and_backward(N):
    from gaussianmesh_amd.model_ops import mesh_activate
    d = _inputs(N, N)
    out = mesh_activate(d["bc"], d["dist"], d["scaling"], d["rot"], d["opac"], d["v1"], d["v2"], d["v3"], d["n"], d["r"], 4.0)
    ref = _torch_ref(d)
    for o, r_ in zip(out, ref):
        assert o.shape == r_.shape and float((o.double() - r_).abs().max()) <= 2e-6 * max(1.0, float(r_.abs().max()))
    ws = [torch.randn_like(o) for o in out]
    leaves = [d[k] for k in ("bc", "dist", "scaling", "rot", "opac")]
    g = torch.autograd.grad(sum((o * w).sum() for o, w in zip(out, ws)), leaves, retain_graph=True)
    gr = torch.autograd.grad(sum((r_ * w.double()).sum() for r_, w in zip(ref, ws)), leaves)
    for a, b, k in zip(g, gr, ("bc", "dist", "scaling", "rot", "opac")):
        assert float((a.double() - b).abs().max()) <= 1e-5 * max(1e-6, float(b.abs().max())), k
    # a missing upstream gradient is a zero gradient
    g2 = torch.autograd.grad((out[0] * ws[0]).sum(), leaves, allow_unused=True)
    assert float(g2[2].abs().max()) == 0.0 and float(g2[0].abs().max()) > 0


def test_mesh_restrict_term_fused_into_the_activation():
    """5th output of mesh_activate == loss.mesh_restrict_loss on the activated scales (value and gradient)."""
    from gaussianmesh_amd.model_ops import mesh_activate
    from gaussianmesh_amd.loss import mesh_restrict_loss
    d = _inputs(5000, 7)
    with torch.no_grad():
        d["scaling"] += 2.0                                            # make a good share of the terms positive
    args = (d["bc"], d["dist"], d["scaling"], d["rot"], d["opac"], d["v1"], d["v2"], d["v3"], d["n"], d["r"], 4.0)
    xyz, sc, rt, op, mr = mesh_activate(*args, mr_weight=0.7)
    ref = mesh_restrict_loss(torch.exp(d["scaling"].double()), d["v1"].double(), d["v2"].double(), d["v3"].double(), weight=0.7)
    assert float(ref) > 0 and abs(float(mr) - float(ref)) <= 1e-5 * float(ref)
    w = torch.randn_like(sc)
    g, = torch.autograd.grad(2.5 * mr + (sc * w).sum(), [d["scaling"]])
    gr, = torch.autograd.grad(2.5 * ref + (torch.exp(d["scaling"].double()) * w.double()).sum(), [d["scaling"]])
    assert float((g.double() - gr).abs().max()) <= 1e-5 * float(gr.abs().max())
    assert len(mesh_activate(*args)) == 4


def test_fused_adam_matches_torch_adam_and_two_rate_tensor():
    from gaussianmesh_amd.model_ops import FusedAdam
    rng = np.random.default_rng(0)
    shapes = [(1000, 3), (1000, 1), (1000, 16, 3), (37,), (5, 4)]
    ps = [torch.tensor(rng.standard_normal(s).astype(np.float32), device="cuda", requires_grad=True) for s in shapes]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    q_dc = qs[2].detach()[:, :1].clone().requires_grad_(True); q_rest = qs[2].detach()[:, 1:].clone().requires_grad_(True)
    lrs = [0.01, 0.02, 0.005, 0.1, 0.03]
    groups = [{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))]
    groups[2].update(lr_rest=0.005 / 20, period=48, split=3)
    opt = FusedAdam(groups, eps=1e-15)
    ref = torch.optim.Adam([{"params": [qs[0]], "lr": lrs[0]}, {"params": [qs[1]], "lr": lrs[1]}, {"params": [q_dc], "lr": lrs[2]},
                            {"params": [q_rest], "lr": lrs[2] / 20}, {"params": [qs[3]], "lr": lrs[3]}, {"params": [qs[4]], "lr": lrs[4]}],
                           lr=0.0, eps=1e-15)
    for it in range(6):
        for p in ps:
            p.grad = torch.tensor(rng.standard_normal(tuple(p.shape)).astype(np.float32), device="cuda")
        qs[0].grad, qs[1].grad, qs[3].grad, qs[4].grad = ps[0].grad.clone(), ps[1].grad.clone(), ps[3].grad.clone(), ps[4].grad.clone()
        q_dc.grad, q_rest.grad = ps[2].grad[:, :1].clone(), ps[2].grad[:, 1:].clone()
        opt.step(); ref.step()
    chk = lambda a, b: float((a.detach() - b.detach()).abs().max()) <= 2e-6
    assert chk(ps[0], qs[0]) and chk(ps[1], qs[1]) and chk(ps[3], qs[3]) and chk(ps[4], qs[4])
    assert chk(ps[2][:, :1], q_dc) and chk(ps[2][:, 1:], q_rest)
    opt.zero_grad()
    assert all(p.grad is None for p in ps)
    ps[0].grad = None
    opt.step()                                             # nothing to do: no gradients


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_fused_adam_active_coefficients_only(deg):
    """gm_adam_step_active: while only the coefficients of SH degrees <= deg have ever had a gradient, an optimizer that skips the
    rest gives, bit for bit, what the full update gives on zero gradients - parameters, both moments - also across the step at
    which the degree goes up."""
    from gaussianmesh_amd.model_ops import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(deg)
    N = 1537
    p0 = torch.randn((N, 16, 3), device="cuda", generator=g)
    other0 = torch.randn((N, 3), device="cuda", generator=g)

    def make():
        p, o = p0.clone().requires_grad_(True), other0.clone().requires_grad_(True)
        return p, o, FusedAdam([{"params": [p], "lr": 0.0025, "lr_rest": 0.0025 / 20, "period": 48, "split": 3, "name": "f"},
                                {"params": [o], "lr": 0.01, "name": "o"}], eps=1e-15)
    pa, oa, full = make()
    pb, ob, lim = make()
    for it in range(7):
        d = deg if it < 4 else deg + 1                                   # the degree goes up before the fifth step
        nc = (d + 1) ** 2
        grad = torch.zeros((N, 16, 3), device="cuda")
        grad[:, :nc] = torch.randn((N, nc, 3), device="cuda", generator=g)
        go = torch.randn((N, 3), device="cuda", generator=g)
        pa.grad, oa.grad, pb.grad, ob.grad = grad.clone(), go.clone(), grad.clone(), go.clone()
        lim.param_groups[0]["active"] = 3 * nc if d < 3 else 0
        full.step(); lim.step()
        assert torch.equal(pa, pb) and torch.equal(oa, ob), it
        for k in ("m", "values"):
            assert torch.equal(full.param_groups[0][k][0], lim.param_groups[0][k][0]), (it, k)
    assert torch.equal(pa.detach()[:, (deg + 2) ** 2:], p0[:, (deg + 2) ** 2:]) or deg == 2       # never-active coefficients never moved
    assert not torch.equal(pa.detach()[:, :1], p0[:, :1])
