"""-m gpu: gm_backward_sh_step at the C-ABI level (round 6) - the backward pass that also takes the Adam step of the SH rows.

What it has to equal is spelled out in numpy from the ORACLE's dL/dSH: m' = b1 m + (1 - b1) g, v' = b2 v + (1 - b2) g^2,
p' = p - lr sqrt(1 - b2^t) / (1 - b1^t) m' / (sqrt(v') + eps) with lr_dc for coefficient 0 and lr_rest for the others (the reference's
"f_dc" / "f_rest" groups, mesh_based_gaussian_model.py:120-131 + jittor's Adam), for the first `rows` rows; coefficients above the active
degree and the rows behind `rows` (a frozen cloud sharing the operand) keep parameter and moments bit for bit; every other gradient is the
ordinary backward's.  Through the Trainer the same is asserted against FusedAdam in tests/test_gpu_train.py."""
import numpy as np
import pytest
import torch

from gpu_utils import T
from helpers import small_scene

pytestmark = pytest.mark.gpu


def _forward(sc, cam, bg, D, shs):
    from gaussianmesh_amd import rasterizer as R
    # the forward of a training step (GM_FWD_EXACT_EXPONENT), as the autograd operator runs it when a gradient is required
    return R.rasterize_forward_begin(T(bg), T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, T(cam["view"]), T(cam["proj"]),
                                     cam["tanx"], cam["tany"], cam["H"], cam["W"], shs, D, T(cam["campos"]), False, False).finish(exact_exponent=True)


def _backward(sc, cam, bg, D, shs, fw, dpix, **kw):
    from gaussianmesh_amd import rasterizer as R
    nr, _, radii, geom, binning, img = fw
    return R.rasterize_backward(T(bg), T(sc["means"]), radii, None, T(sc["scales"]), T(sc["rots"]), 1.0, None, T(cam["view"]), T(cam["proj"]), cam["tanx"],
                                cam["tany"], T(dpix), shs, D, T(cam["campos"]), geom, nr, binning, img, False, **kw)


@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_backward_with_the_sh_rows_adam_step_vs_oracle_gradient_and_numpy_adam(D, oracle):
    from gaussianmesh_amd import rasterizer as R
    P, rows = 1800, 1301                                    # the last 499 rows are "frozen": behind the optimizer's rows
    sc, cam = small_scene(P=P, W=176, H=112, seed=11 + D, D=3)
    bg = np.array([0.2, 0.4, 0.1], np.float32)
    dpix = np.random.default_rng(5).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    rng = np.random.default_rng(6)
    m0 = (1e-3 * rng.normal(size=(rows, 16, 3))).astype(np.float32)
    v0 = (1e-6 * rng.uniform(0.1, 1.0, size=(rows, 16, 3))).astype(np.float32)
    nq = (D + 1) ** 2
    m0[:, nq:] = 0.0; v0[:, nq:] = 0.0                      # (coefficients that never had a gradient: moments at rest)
    lr_dc, lr_rest, betas, eps, step = 2.5e-3, 1.25e-4, (0.9, 0.999), 1e-15, 7
    # the ordinary pass first: gradients to compare with, and the oracle's dL/dSH for the numpy step
    shs = T(sc["shs"])
    fw = _forward(sc, cam, bg, D, shs)
    plain = _backward(sc, cam, bg, D, shs, fw, dpix)
    fwo = oracle.forward_full(sc, cam, bg, D=D)
    bwo = oracle.backward_full(sc, cam, bg, fwo, dpix, D=D)
    g = np.asarray(bwo["dsh"], np.float64)[:rows]
    got_g = plain[5].cpu().numpy().astype(np.float64)[:rows]
    assert np.abs(got_g - g).max() <= 1e-3 * np.abs(g).max()
    # the fused pass, from the same forward state
    p_dev = T(sc["shs"])
    m_dev, v_dev = T(m0), T(v0)
    step_obj = R.ShStep(p_dev[:rows], m_dev, v_dev, lr_dc, lr_rest, betas, eps, step)
    fw2 = _forward(sc, cam, bg, D, p_dev)
    fused = _backward(sc, cam, bg, D, p_dev, fw2, dpix, sh_step=step_obj)
    torch.cuda.synchronize()
    assert step_obj.applied and fused[1] is None and fused[5] is None
    for k in (0, 2, 3, 6, 7):                                # dmeans2D, dopacity, dmeans3D, dscales, drots: the ordinary backward's
        a, b = fused[k].cpu().numpy().astype(np.float64), plain[k].cpu().numpy().astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30), k       # (float atomics in another order, nothing else)
    # the moments: linear / quadratic in the gradient - numpy on the DEVICE's own dL/dSH (the ordinary pass's output) to float32 rounding
    # and atomics order, on the ORACLE's to the gradient gate
    b1, b2 = betas
    corr = np.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    lr = np.full((1, 16, 1), lr_rest); lr[:, 0] = lr_dc
    p0 = sc["shs"].astype(np.float64)[:rows]
    p_got, m_got, v_got = (t.cpu().numpy() for t in (p_dev, m_dev, v_dev))
    act = slice(0, nq)
    for grad, tol in ((got_g, 2e-5), (g, 2e-3)):
        m1 = b1 * m0.astype(np.float64) + (1 - b1) * grad
        v1 = b2 * v0.astype(np.float64) + (1 - b2) * grad * grad
        assert np.abs(m_got[:, act] - m1[:, act]).max() <= tol * np.abs(m1[:, act]).max(), (tol, "exp_avg")
        assert np.abs(v_got[:, act] - v1[:, act]).max() <= 2 * tol * np.abs(v1[:, act]).max(), (tol, "exp_avg_sq")
    # the parameter: the rule applied to the moments the device wrote (so that this line checks the rule, not the gradient again)
    m64, v64 = m_got.astype(np.float64), v_got.astype(np.float64)
    p1 = p0 - lr * corr * m64 / (np.sqrt(v64) + eps)
    stepsize = np.abs(p1 - p0)[:, act]
    err = np.abs(p_got[:rows, act].astype(np.float64) - p1[:, act])
    assert (err <= 2e-6 * stepsize + 1.3e-7 * np.abs(p0[:, act]) + 1e-12).all(), float(err.max())
    assert stepsize.max() > 0.1 * lr_rest                                         # (steps of the expected size were taken)
    # what the rule leaves alone is untouched bit for bit: coefficients above the degree, rows behind `rows`
    assert np.array_equal(p_got[:rows, nq:], sc["shs"][:rows, nq:]) and np.array_equal(m_got[:, nq:], m0[:, nq:]) and np.array_equal(v_got[:, nq:], v0[:, nq:])
    assert np.array_equal(p_got[rows:], sc["shs"][rows:])
    assert np.abs(p_got[:rows, act] - sc["shs"][:rows, act]).max() > 0.0


def test_sh_step_argument_errors():
    from gaussianmesh_amd import _lib, rasterizer as R
    P = 600
    sc, cam = small_scene(P=P, W=96, H=64, seed=2, D=3)
    bg = np.zeros(3, np.float32)
    dpix = np.ones((3, cam["H"], cam["W"]), np.float32)
    shs = T(sc["shs"])
    fw = _forward(sc, cam, bg, 3, shs)
    keep = shs.clone()
    m, v = torch.zeros_like(shs), torch.zeros_like(shs)
    with pytest.raises(ValueError):                                               # moments of another shape
        R.ShStep(shs, m[:10], v, 1e-3, 1e-4, (0.9, 0.999), 1e-15, 1)
    with pytest.raises(_lib.GmeshError, match="shs operand"):                     # a step for another tensor than this pass's operand
        _backward(sc, cam, bg, 3, shs, fw, dpix, sh_step=R.ShStep(shs.clone(), m, v, 1e-3, 1e-4, (0.9, 0.999), 1e-15, 1))
    with pytest.raises(_lib.GmeshError, match="optimizer state"):                 # step 0: no bias correction exists
        _backward(sc, cam, bg, 3, shs, fw, dpix, sh_step=R.ShStep(shs, m, v, 1e-3, 1e-4, (0.9, 0.999), 1e-15, 0))
    lib = _lib.lib()
    rc = lib.gm_backward_sh_step(2, P, 3, 15, 1, None, cam["W"], cam["H"], None, shs.data_ptr(), None, 1.0, None, None, None, None, None, 1.0, 1.0, None, None, None, None,
                                 None, None, None, None, None, None, None, P, m.data_ptr(), v.data_ptr(), 1e-3, 1e-4, 0.9, 0.999, 1e-15, 1, 0, None)
    assert rc == 1                                                                # GM_ERR_INVALID_ARG: M != 16
    torch.cuda.synchronize()
    assert torch.equal(shs, keep) and not m.any() and not v.any()                 # nothing was stepped by a refused call
