"""One-off fuzz of the emission policies (run on a GPU box): 40 random scenes, every policy vs the reference emission."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gaussianmesh_amd import scenes, _lib
from gpu_utils import forward_state
bad = 0
for seed in range(int(os.environ.get("FUZZ_SEEDS", "40"))):
    rng = np.random.default_rng(seed)
    P = int(rng.integers(200, 30000))
    lo = float(10 ** rng.uniform(-3, -1)); hi = lo * float(10 ** rng.uniform(0.3, 2.2))
    sc = scenes.make_cloud(P, seed=seed, scale_lo=lo, scale_hi=hi)
    if seed % 3 == 0:                                   # needle-shaped splats
        sc["scales"][:, 0] *= 20.0
    if seed % 4 == 1:
        sc["opac"][:] = rng.uniform(0.003, 0.02, size=sc["opac"].shape).astype(np.float32)     # around the 1/255 threshold
    W = int(rng.integers(17, 700)); H = int(rng.integers(17, 500))
    cam = scenes.orbit_camera(int(rng.integers(0, 16)), 16, W, H, radius=float(rng.uniform(0.5, 9.0)))
    bg = rng.random(3).astype(np.float32)
    ref = forward_state(sc, cam, bg, D=3, tile_cull=0)
    for mode in (1, 2, 3):
        cu = forward_state(sc, cam, bg, D=3, tile_cull=mode)
        ok = np.array_equal(cu["color"], ref["color"]) and np.array_equal(cu["final_T"], ref["final_T"]) and np.array_equal(cu["radii"], ref["radii"])
        if not ok:
            bad += 1
            d = np.abs(cu["color"] - ref["color"])
            print("MISMATCH seed", seed, "mode", mode, "P", P, W, H, "max", d.max(), "npix", (d > 0).sum(), "R", ref["R"], cu["R"])
    print("seed", seed, "P", P, "%dx%d" % (W, H), "scales %.3g..%.3g" % (lo, hi), "R", ref["R"], flush=True)
print("mismatches:", bad)
