#!/usr/bin/env python
"""Repeats tests/test_gpu_bench_multirank.py's two-rank run and says HOW a rank's last image differs from the single-process render of
the same frame (pixels, magnitude, and which animation frame it matches instead, if any)."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_bench_multirank import _launch
import bench
from gpu_utils import T
from gaussianmesh_amd import multiview, rasterizer as Rz, scenes
from gaussianmesh_amd.deform import mesh_rs, pack_mesh_state
P, W, H, F, steps, warm = 20000, 320, 200, 8, 4, 2
host = bench.build_scene(P, W, H, F)
g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
g["tri"] = T(host["tri"], dtype=torch.int32)
faces = T(host["faces"], dtype=torch.int32)
def render(frame, view):
    cam = scenes.orbit_camera(view, F, W, H)
    ct = {n: T(cam[n]) for n in ("view", "proj", "campos")}
    state = mesh_rs(g["verts"], T(host["mesh"][frame][:, 0:3]), faces, want_state=True)[2]
    packed = pack_mesh_state(state, g["verts"])
    return Rz.forward_deformed_begin(torch.ones(3, device="cuda"), g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], ct["view"],
                                     ct["proj"], cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish()[1].cpu().numpy()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
last = warm + steps - 1
for k in range(runs):
    tmp = pathlib.Path(tempfile.mkdtemp())
    out = _launch(tmp, 2, dict(GM_BENCH_SHARE_DEVICE="1", GM_BENCH_BACKEND="gloo"), P, W, H, F, steps, warm, batch)
    for r in range(2):
        d = np.load(os.path.join(str(tmp), "rank%d.npz" % r))
        v = multiview.view_for_step(last, F, r, 2)
        ref = render(last % F, v)
        if not np.array_equal(d["image"], ref):
            diff = np.abs(d["image"] - ref)
            msg = "run %d rank %d: %d pixels differ, max %.3g, overflows %d" % (k, r, int((diff > 0).any(axis=0).sum()), diff.max(), int(d["overflows"]))
            for t in range(F):
                for vv in range(F):
                    if np.array_equal(d["image"], render(t, vv)):
                        msg += " | equals frame %d view %d (expected frame %d view %d)" % (t, vv, last % F, v)
            print(msg, flush=True)
print("done")
