"""GPU parity: gm_ssim_fwd / gm_ssim_bwd (through gaussianmesh_amd.loss) against oracle/loss_oracle.py.
Tolerances: loss values 2e-6 absolute (float32 separable sums vs float64 2-D sums), gradients 1e-3 relative to the
largest gradient magnitude (north_star's gradient bar)."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

pytestmark = pytest.mark.gpu


def _imgs(seed, shape):
    rng = np.random.default_rng(seed)
    a = rng.random(shape).astype(np.float32)
    b = np.clip(a + 0.15 * rng.standard_normal(shape), 0, 1).astype(np.float32)
    return a, b


@pytest.mark.parametrize("shape", [(3, 64, 64), (3, 37, 45), (1, 5, 7), (3, 130, 97), (2, 3, 40, 33)])
def test_ssim_forward_and_gradient(shape):
    from gaussianmesh_amd import loss
    a, b = _imgs(0, shape)
    ta = torch.tensor(a, device="cuda", requires_grad=True)
    tb = torch.tensor(b, device="cuda")
    s = loss.ssim(ta, tb)
    oa = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    so = lo.ssim_torch(oa, torch.tensor(b))
    assert abs(float(s) - float(so)) < 2e-6
    s.backward()
    go, = torch.autograd.grad(so, oa)
    g = ta.grad.cpu().double()
    assert float((g - go).abs().max()) <= 1e-3 * float(go.abs().max())


def test_ssim_per_image_means():
    from gaussianmesh_amd import loss
    a, b = _imgs(1, (3, 3, 48, 40))
    ta = torch.tensor(a, device="cuda", requires_grad=True)
    s = loss.ssim(ta, torch.tensor(b, device="cuda"), size_average=False)
    oa = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    so = lo.ssim_torch(oa, torch.tensor(b), size_average=False)
    assert s.shape == (3,) and float((s.cpu().double() - so).abs().max()) < 2e-6
    w = torch.tensor([1.0, -2.0, 0.5])
    (s * w.cuda()).sum().backward()
    go, = torch.autograd.grad((so * w.double()).sum(), oa)
    assert float((ta.grad.cpu().double() - go).abs().max()) <= 1e-3 * float(go.abs().max())


@pytest.mark.parametrize("lam", [0.2, 0.0, 1.0])
def test_photometric_loss_matches_oracle(lam):
    from gaussianmesh_amd import loss
    a, b = _imgs(2, (3, 90, 120))
    ta = torch.tensor(a, device="cuda", requires_grad=True)
    tb = torch.tensor(b, device="cuda")
    L = loss.photometric_loss(ta, tb, lam)
    oa = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    Lo = lo.photometric_torch(oa, torch.tensor(b), lam)
    assert abs(float(L) - float(Lo)) < 2e-6
    (3.0 * L).backward()
    go, = torch.autograd.grad(3.0 * Lo, oa)
    assert float((ta.grad.cpu().double() - go).abs().max()) <= 1e-3 * float(go.abs().max())
    # the unfused composition of the reference's training loop gives the same value
    L2 = (1.0 - lam) * loss.l1_loss(ta.detach(), tb) + lam * (1.0 - loss.ssim(ta.detach(), tb))
    assert abs(float(L2) - float(L)) < 1e-6


def test_ssim_known_answers_and_errors():
    from gaussianmesh_amd import loss, _lib
    a, _ = _imgs(3, (3, 33, 65))
    ta = torch.tensor(a, device="cuda")
    assert abs(float(loss.ssim(ta, ta)) - 1.0) < 1e-6
    assert float(loss.photometric_loss(ta, ta, 0.2)) < 1e-6
    with pytest.raises(ValueError):
        loss.ssim(ta, ta[:, :-1])
    with pytest.raises(NotImplementedError):
        loss.ssim(ta, ta, window_size=7)
    with pytest.raises(_lib.GmeshError):
        loss.ssim(ta.cpu(), ta.cpu())


def test_full_hd_loss_properties():
    """1080p: value against the torch float64 evaluation, and a directional finite difference of the fused loss."""
    from gaussianmesh_amd import loss
    a, b = _imgs(4, (3, 1080, 1920))
    ta = torch.tensor(a, device="cuda", requires_grad=True)
    tb = torch.tensor(b, device="cuda")
    L = loss.photometric_loss(ta, tb, 0.2)
    Lo = lo.photometric_torch(torch.tensor(a), torch.tensor(b), 0.2)
    assert abs(float(L) - float(Lo)) < 2e-6
    L.backward()
    g = ta.grad
    d = torch.sign(g) * 1e-3
    Lp = loss.photometric_loss((ta.detach() + d), tb, 0.2)
    pred = float((g.double() * d.double()).sum())
    assert abs((float(Lp) - float(L)) - pred) < 0.05 * abs(pred) + 1e-6
