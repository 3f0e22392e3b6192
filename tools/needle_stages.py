#!/usr/bin/env python
"""Where does the float32 error of a needle scene's gradients come from?  CPU only (C oracle vs oracle/torch_dense.py, float64 autograd).
Stage split of backward.cu: the BLEND backward (per-pixel alpha from a cancelling quadratic form, float sums over pixels -> dL/dconic,
dL/dmean2D) and the PREPROCESS backward (conic -> cov2D -> cov3D -> scale / rotation, all per Gaussian).  The float64 truth of the
interface (dL/dconic, dL/dmean2D) is fed into the float32 preprocess backward to time the second stage alone.

    python tools/needle_stages.py 120 168
"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import oracle, torch_dense as td
from helpers import fuzz_scene

t64 = lambda a, rg=False: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)
rel = lambda a, b: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b, np.float64)).max(), 1e-30)
for seed in [int(a) for a in sys.argv[1:]] or [120]:
    sc, cam, bg, D, pre_cov, pre_col, dpix = fuzz_scene(seed)
    W, H = cam["W"], cam["H"]
    fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
    means, opac = t64(sc["means"], True), t64(sc["opac"], True)
    kw, leaves = {}, {"means": means}
    if pre_col: kw["colors_precomp"] = t64(sc["colors_precomp"], True)
    else: kw["shs"] = t64(sc["shs"], True)
    if pre_cov: kw["cov3D_precomp"] = leaves["cov"] = t64(sc["cov3D_precomp"], True)
    else:
        kw["scales"] = leaves["scales"] = t64(sc["scales"], True); kw["rots"] = leaves["rots"] = t64(sc["rots"], True)
    out, aux = td.render(means, opac, t64(cam["view"]), t64(cam["proj"]), t64(cam["campos"]), W, H, cam["tanx"], cam["tany"], t64(bg), D=D, **kw)
    live = aux["live"]
    for v in live.values(): v.retain_grad()
    (out * t64(dpix)).sum().backward()
    dconic64 = live["conic"].grad.numpy()                                        # [P, 3]: d/d(conic.x, conic.y, conic.z)
    dm2d64 = np.stack([live["px"].grad.numpy() * 0.5 * W, live["py"].grad.numpy() * 0.5 * H], 1)
    dconic32 = bw["dconic"][:, [0, 1, 3]] * np.array([1.0, 2.0, 1.0])       # (backward.cu:549-551 accumulates HALF the off-diagonal derivative)
    print("seed %d: blend stage, float32 oracle vs float64: dL/dconic %.2e  dL/dmean2D %.2e" % (seed, rel(dconic32, dconic64), rel(bw["dmean2D"][:, :2], dm2d64)))
    # second stage alone: the float64 interface gradients through the float32 preprocess backward
    dc = np.zeros_like(bw["dconic"]); dc[:, 0] = dconic64[:, 0]; dc[:, 1] = 0.5 * dconic64[:, 1]; dc[:, 3] = dconic64[:, 2]
    dm = np.zeros_like(bw["dmean2D"]); dm[:, :2] = dm2d64
    r = oracle.preprocess_bwd(sc["means"], fw["geo"], cam["view"], cam["proj"], cam["campos"], W, H, cam["tanx"], cam["tany"], dm, dc, bw["dcolor"], D=D,
                              shs=None if pre_col else sc["shs"], scales=None if pre_cov else sc["scales"], rots=None if pre_cov else sc["rots"])
    names = dict(means=(bw["dmean3D"], r[0]), cov=(bw["dcov3D"], r[1]), scales=(bw["dscale"], r[3]), rots=(bw["drot"], r[4]))
    for k, leaf in leaves.items():
        truth = leaf.grad.numpy(); full, second = names[k]
        print("   d/d%-7s whole float32 chain %.2e   float64 blend + float32 preprocess backward %.2e" % (k, rel(full.reshape(truth.shape), truth), rel(second.reshape(truth.shape), truth)))
