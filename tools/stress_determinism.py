#!/usr/bin/env python
"""Stress: the same frame rendered N times through the exact-count path (forward_deformed_begin().finish() and rasterize_forward);
every image must be bit-identical to the first and every instance count equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench
from gpu_utils import T
from gaussianmesh_amd import rasterizer as Rz, scenes
from gaussianmesh_amd.deform import mesh_rs, pack_mesh_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
P, W, H, F = 20000, 320, 200, 8
host = bench.build_scene(P, W, H, F)
g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
g["tri"] = T(host["tri"], dtype=torch.int32)
faces = T(host["faces"], dtype=torch.int32)
cam = scenes.orbit_camera(3, F, W, H)
ct = {n: T(cam[n]) for n in ("view", "proj", "campos")}
bg = torch.ones(3, device="cuda")
ref = None
bad = 0
for it in range(N):
    state = mesh_rs(g["verts"], T(host["mesh"][it % F][:, 0:3]) if it % 50 == 0 else v1, faces, want_state=True)[2] if it % 50 == 0 or True else None
    if it == 0:
        v1 = T(host["mesh"][5][:, 0:3])
        state = mesh_rs(g["verts"], v1, faces, want_state=True)[2]
    packed = pack_mesh_state(mesh_rs(g["verts"], v1, faces, want_state=True)[2], g["verts"])
    nr, color, radii, *_ = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], ct["view"], ct["proj"],
                                                     cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish()
    img = color.cpu().numpy()
    if ref is None:
        ref, nref, rref = img, nr, radii.cpu().numpy()
    elif nr != nref or not np.array_equal(img, ref):
        bad += 1
        r = radii.cpu().numpy()
        print("iteration", it, "differs: num_rendered", nr, "vs", nref, "| pixels differing", int((img != ref).any(axis=0).sum()), "| radii differing", int((r != rref).sum()),
              "| image min/max", img.min(), img.max(), flush=True)
print("frames", N, "mismatches", bad, "num_rendered", nref)
