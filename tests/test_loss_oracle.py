"""CPU: the loss oracle's two independent evaluations agree; known answers."""
import numpy as np
import torch

from oracle import loss_oracle as lo


def _imgs(seed, C=3, H=37, W=45):
    rng = np.random.default_rng(seed)
    a = rng.random((C, H, W)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal((C, H, W)), 0, 1).astype(np.float32)
    return a, b


def test_window_is_normalised_float32():
    g = lo.window_1d()
    assert g.dtype == np.float32 and g.shape == (11,)
    assert abs(float(g.sum(dtype=np.float64)) - 1.0) < 1e-6
    assert np.array_equal(g, g[::-1]) and g.argmax() == 5
    w = lo.window_2d()
    assert w.shape == (11, 11) and np.array_equal(w, w.T)


def test_scipy_and_torch_evaluations_agree():
    a, b = _imgs(0)
    m = lo.ssim_map(a, b)
    t = lo.ssim_torch(torch.tensor(a), torch.tensor(b))
    assert abs(m.mean() - float(t)) < 1e-12
    per = lo.ssim_torch(torch.tensor(np.stack([a, b])), torch.tensor(np.stack([b, b])), size_average=False)
    assert abs(float(per[0]) - m.mean()) < 1e-12 and abs(float(per[1]) - 1.0) < 1e-12


def test_known_answers():
    a, b = _imgs(1)
    assert abs(lo.ssim(a, a) - 1.0) < 1e-12                   # identical images
    assert abs(lo.ssim(a, b) - lo.ssim(b, a)) < 1e-12         # symmetric
    assert lo.ssim(a, b) < 1.0
    c = np.full((1, 30, 30), 0.5, np.float32)
    # constant images x = 0.5, y = 0.25: in the interior every window sum is value * S, S = sum of the window (1 + 8e-8)
    m = lo.ssim_map(c, 0.25 * c / 0.5)
    S = float(lo.window_2d().astype(np.float64).sum())
    mu1, mu2 = 0.5 * S, 0.25 * S
    s1, s2, s12 = 0.25 * S - mu1 ** 2, 0.0625 * S - mu2 ** 2, 0.125 * S - mu1 * mu2
    want = (2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))
    assert abs(m[0, 15, 15] - want) < 1e-12
    assert abs(lo.l1(a, b) - np.abs(a.astype(np.float64) - b).mean()) < 1e-15


def test_gradient_of_torch_evaluation_by_finite_differences():
    a, b = _imgs(2, C=1, H=16, W=18)
    ta = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    loss = lo.photometric_torch(ta, torch.tensor(b), 0.2)
    g, = torch.autograd.grad(loss, ta)
    rng = np.random.default_rng(3)
    for _ in range(5):
        c, y, x = 0, int(rng.integers(16)), int(rng.integers(18))
        e = 1e-6
        ap, am = a.astype(np.float64).copy(), a.astype(np.float64).copy()
        ap[c, y, x] += e; am[c, y, x] -= e
        f = lambda z: 0.8 * np.abs(z - b).mean() + 0.2 * (1 - lo.ssim_map(z, b).mean())
        fd = (f(ap) - f(am)) / (2 * e)
        assert abs(fd - float(g[c, y, x])) < 1e-6 * max(1.0, abs(fd))
