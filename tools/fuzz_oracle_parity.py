"""One-off fuzz (run on a GPU box): forward image and all gradients against the CPU oracle on random small scenes."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gaussianmesh_amd import scenes
from oracle import oracle
from test_gpu_parity import _grads_gpu, _rel
from helpers import fuzz_scene
worst_f, worst_g, bad = 0.0, 0.0, 0
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for seed in range(NSEEDS):
    sc, cam, bg, D, pre_cov, pre_col, dpix = fuzz_scene(seed); P, W, H = sc["means"].shape[0], cam["W"], cam["H"]
    fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
    err = np.abs(color - fw["color"])
    f_out = (err > 1e-4).mean()
    pairs = [("means", bw["dmean3D"]), ("opac", bw["dopacity"])]
    pairs += [("colors", bw["dcolor"])] if pre_col else [("shs", bw["dsh"])]
    pairs += [("cov", bw["dcov3D"])] if pre_cov else [("scales", bw["dscale"]), ("rots", bw["drot"])]
    rels = {k: _rel(np.asarray(g[k]).reshape(np.asarray(r).shape), r) for k, r in pairs}
    ok = np.array_equal(radii, fw["geo"]["radii"]) and f_out <= 1e-4 and err.max() <= 5e-3 and max(rels.values()) <= 1e-3
    worst_f = max(worst_f, err.max()); worst_g = max(worst_g, max(rels.values()))
    bad += (not ok)
    print("seed", seed, "P", P, "%dx%d" % (W, H), "D", D, "precomp", pre_cov, pre_col, "fwd max %.2e (>1e-4: %.1e)" % (err.max(), f_out),
          "grad rel max %.2e" % max(rels.values()), "" if ok else "  <-- FAIL " + str(rels), flush=True)
print("failures:", bad, "worst fwd %.2e worst grad rel %.2e" % (worst_f, worst_g))
