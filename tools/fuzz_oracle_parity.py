"""One-off fuzz (run on a GPU box): forward image and all gradients against the CPU oracle on random small scenes."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gaussianmesh_amd import scenes
from oracle import oracle
from test_gpu_parity import _grads_gpu, _rel
worst_f, worst_g, bad = 0.0, 0.0, 0
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for seed in range(NSEEDS):
    rng = np.random.default_rng(100 + seed)
    P = int(rng.integers(50, 3000))
    lo = float(10 ** rng.uniform(-2.3, -1)); hi = lo * float(10 ** rng.uniform(0.3, 1.6))
    sc = scenes.make_cloud(P, seed=seed, scale_lo=lo, scale_hi=hi)
    if seed % 3 == 0:
        sc["scales"][:, 0] *= 10.0
    W = int(rng.integers(17, 160)); H = int(rng.integers(17, 120))
    cam = scenes.orbit_camera(int(rng.integers(0, 16)), 16, W, H, radius=float(rng.uniform(2.0, 9.0)))
    bg = rng.random(3).astype(np.float32)
    D = int(rng.integers(0, 4))
    pre_cov, pre_col = bool(seed % 2), bool((seed // 2) % 2)
    if pre_cov:
        sc["cov3D_precomp"] = scenes.strip_symmetric(scenes.cov3d_from_scale_rot(sc["scales"], sc["rots"])).astype(np.float32)
    if pre_col:
        sc["colors_precomp"] = rng.random((P, 3)).astype(np.float32)
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
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
