"""Cross-checks oracle/gm_oracle.c against the independent torch float64 dense restatement (oracle/torch_dense.py):
forward image, radii, n_contrib, and EVERY gradient via autograd.  This is what stands in for a reference build
(see the oracle header: parity unpinned for the CUDA core)."""
import numpy as np
import pytest
import torch

from helpers import small_scene


def _T(a, rg=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("mode", ["sh_scale_rot", "precomp"])
@pytest.mark.parametrize("seed,D,P,W,H", [(0, 3, 160, 40, 36), (4, 1, 160, 40, 36), (9, 3, 2400, 128, 96)])
def test_oracle_vs_dense_autograd(oracle, mode, seed, D, P, W, H):
    """The third case is large enough for the tile machinery to matter: 2400 Gaussians on 48 tiles, lists of several hundred
    entries (more than one 256-entry staging round of RAST/forward.cu:306-330 per tile), pixels that saturate and stop early."""
    from oracle import torch_dense as td
    sc, cam = small_scene(P=P, W=W, H=H, seed=seed, D=D, scale_lo=0.05, scale_hi=0.6 if P < 1000 else 0.35)
    bg = np.array([0.2, 0.5, 0.9], np.float32)
    pre = mode == "precomp"
    fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre, use_precomp_color=pre)
    means, opac = _T(sc["means"], True), _T(sc["opac"], True)
    m2d = torch.zeros(means.shape[0], 3, dtype=torch.float64, requires_grad=True)
    kw = {}
    if pre:
        cols, cov = _T(sc["colors_precomp"], True), _T(sc["cov3D_precomp"], True)
        kw = dict(colors_precomp=cols, cov3D_precomp=cov)
    else:
        shs, scales, rots = _T(sc["shs"], True), _T(sc["scales"], True), _T(sc["rots"], True)
        kw = dict(shs=shs, scales=scales, rots=rots)
    out, aux = td.render(means, opac, _T(cam["view"]), _T(cam["proj"]), _T(cam["campos"]), cam["W"], cam["H"], cam["tanx"],
                         cam["tany"], _T(bg), D=D, means2D=m2d, **kw)
    if P > 1000:
        r = fw["bins"]["ranges"].astype(np.int64)
        assert (r[:, 1] - r[:, 0]).max() > 256 and fw["n_contrib"].max() > 256          # multi-round tiles, deep contributors
        assert (fw["final_T"] < 1e-3).any()                                                # some pixels saturate (early stop)
    assert np.array_equal(aux["radii"].numpy(), fw["geo"]["radii"])
    assert np.array_equal(aux["n_contrib"].numpy(), fw["n_contrib"].astype(np.int64))
    assert np.abs(out.detach().numpy() - fw["color"]).max() <= 2e-6
    dpix = np.random.default_rng(seed + 1).normal(size=tuple(out.shape)).astype(np.float32)
    (out * _T(dpix)).sum().backward()
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre, use_precomp_color=pre)
    tol = 2e-5
    assert _rel(bw["dmean3D"], means.grad.numpy()) <= tol
    assert _rel(bw["dopacity"], opac.grad.numpy().reshape(-1)) <= tol
    assert _rel(bw["dmean2D"][:, :2], m2d.grad.numpy()[:, :2]) <= tol
    if pre:
        assert _rel(bw["dcolor"], cols.grad.numpy()) <= tol
        assert _rel(bw["dcov3D"], cov.grad.numpy()) <= tol
    else:
        assert _rel(bw["dsh"], shs.grad.numpy()) <= tol
        assert _rel(bw["dscale"], scales.grad.numpy()) <= tol
        assert _rel(bw["drot"], rots.grad.numpy()) <= tol
