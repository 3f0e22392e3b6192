#!/usr/bin/env python
"""Sizes of the depth buckets of one C3 frame (GeomState::bucket_start, counters)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gaussianmesh_amd import _lib, rasterizer as Rz, scenes  # noqa: E402
from gaussianmesh_amd.deform import pack_mesh_state  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_utils import _view  # noqa: E402

P, W, H, F = 1_000_000, 1920, 1080, 64
dev = torch.device("cuda:0")
host = bench.build_scene(P, W, H, F)
g = {k: torch.tensor(host[k], device=dev) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
g["tri"] = torch.tensor(host["tri"], dtype=torch.int32, device=dev)
cam = scenes.orbit_camera(3, F, W, H)
ct = {n: torch.tensor(cam[n], device=dev) for n in ("view", "proj", "campos")}
packed = pack_mesh_state(torch.tensor(host["mesh"][3], device=dev), g["verts"])
lib0 = _lib.lib()
fn = lib0.gm_debug_bucket_trace; fn.restype = None; fn.argtypes = [C.c_void_p]
tbuf = torch.zeros((8192 * 3,), dtype=torch.int64, device=dev)      # bucket sort | depth scatter | tile scatter records
for _ in range(3):
    Rz.forward_deformed_begin(torch.ones(3, device=dev), g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"],
                              g["opac"], ct["view"], ct["proj"], cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish()
torch.cuda.synchronize()
fn(tbuf.data_ptr())
nr, color, radii, geom, binning, img = Rz.forward_deformed_begin(torch.ones(3, device=dev), g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"],
                                                               g["opac"], ct["view"], ct["proj"], cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish()
torch.cuda.synchronize()
lib = _lib.lib()
lib.gm_geom_field.restype = C.c_void_p
bs = _view(geom, lib.gm_geom_field(geom.data_ptr(), P, b"bucket_start"), 2049, torch.int32).astype(np.int64)
cnt = _view(geom, lib.gm_geom_field(geom.data_ptr(), P, b"counters"), 16, torch.int32)
sz = np.diff(bs)
print("counters", cnt, "instances", nr)
print("buckets in use", (sz > 0).sum(), "max", sz.max(), "mean of non-empty", sz[sz > 0].mean(), "over 4096:", (sz > 4096).sum(),
      "percentiles 50/90/99", np.percentile(sz[sz > 0], [50, 90, 99]))
top = np.argsort(-sz)[:12]
print("largest buckets (index: size):", [(int(i), int(sz[i])) for i in top], "first / last used bucket", int(np.nonzero(sz)[0][0]), int(np.nonzero(sz)[0][-1]))

fn(None)
tr = tbuf.cpu().numpy().reshape(-1, 3)[:2048]
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
st, en, nn = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, tr[:, 2]
print("bucket_sort: workgroups with work", len(tr), "span %.1f us" % en.max(), "start percentiles 50/99/max", np.percentile(st, [50, 99, 100]),
      "duration percentiles 50/90/99/max", np.percentile(en - st, [50, 90, 99, 100]))
late = np.argsort(-en)[:8]
print("latest:", [(round(st[i], 1), round(en[i], 1), int(nn[i])) for i in late])
