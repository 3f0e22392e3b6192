#!/usr/bin/env python
"""Where do the multi-wave ordering kernels lose their time inside the 4-stream frame loop?  Per-workgroup start / end
timestamps of bucket_sort_kernel and of the two scatter kernels (gm_debug_bucket_trace: wall_clock64 at entry and exit of every workgroup), one trace buffer per
frame, for the same frames issued (a) one at a time on one stream and (b) pipelined over four streams as bench.py does.
rocprofv3 --pmc cannot answer this: it serialises the dispatches it counts (profiles/r03_4stream_pmc_serialized.txt).
Prints, per mode: kernel span (first start to last end), how late workgroups START relative to the first one (placement), and
how long a workgroup RUNS (resident time)."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gaussianmesh_amd import _lib, rasterizer as Rz, scenes  # noqa: E402
from gaussianmesh_amd.deform import mesh_rs_packed, vertex_face_adjacency  # noqa: E402

P, W, H, F = 1_000_000, 1920, 1080, 64
dev = torch.device("cuda:0")
host = bench.build_scene(P, W, H, F)
g = {k: torch.tensor(host[k], device=dev) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
g["tri"] = torch.tensor(host["tri"], dtype=torch.int32, device=dev)
g["faces"] = torch.tensor(host["faces"], dtype=torch.int32, device=dev)
off, adj = vertex_face_adjacency(g["faces"], g["verts"].shape[0])
adjacency = (torch.tensor(off, device=dev), torch.tensor(adj, device=dev))
v1 = torch.tensor(host["mesh"][:, :, 0:3], device=dev).contiguous()
cams = [scenes.orbit_camera(k, F, W, H) for k in range(F)]
ct = [{n: torch.tensor(c[n], device=dev) for n in ("view", "proj", "campos")} for c in cams]
bg = torch.ones(3, device=dev)
lib = _lib.lib()
fn = lib.gm_debug_bucket_trace; fn.restype = None; fn.argtypes = [C.c_void_p]
hint = Rz.new_work_hint(W, H, dev)


def run(nstreams, nframes, traced):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    ws = [Rz.RasterWorkspace(growth=1.5) for _ in range(nstreams + 8)]
    bufs = {i: torch.zeros((8192 * 3,), dtype=torch.int64, device=dev) for i in traced}
    pend = []
    for i in range(-2 * F, nframes):                       # two passes over the orbit size every workspace, untraced
        with torch.cuda.stream(streams[i % nstreams]):
            fn(bufs[i].data_ptr() if i in bufs else None)
            packed = mesh_rs_packed(g["verts"], v1[i % F], g["faces"], adjacency)
            c = ct[i % F]
            h = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], c["view"], c["proj"],
                                          cams[i % F]["tanx"], cams[i % F]["tany"], H, W, 3, c["campos"], False, workspace=ws[i % len(ws)])
            h.finish(sync_free=i >= -F, image_only=True, work_hint=hint)
            pend.append(h)
            if len(pend) > 6:
                pend.pop(0).check()
    torch.cuda.synchronize()
    fn(None)
    for h in pend:
        h.check()
    out = {}
    for name, lo, hi in (("bucket_sort_kernel", 0, 2048), ("bk_scatter<true> (depth partition)", 2048, 4096), ("bk_scatter<false> (tile pass)", 4096, 8192)):
        rows = []
        for i in traced:
            tr = bufs[i].cpu().numpy().reshape(-1, 3)[lo:hi]
            tr = tr[tr[:, 0] > 0]
            t0 = tr[:, 0].min()
            st, en = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0
            rows.append((en.max(), np.percentile(st, [50, 90, 100]), np.percentile(en - st, [50, 90, 100]), len(tr)))
        out[name] = rows
    return out


for label, ns in (("one stream", 1), ("four streams", 4)):
    res = run(ns, 40, list(range(20, 28)))
    print(label)
    for name, rows in res.items():
        print(" ", name)
        for span, st, du, n in rows:
            print("    span %6.1f us | start after the first workgroup: median %5.1f  p90 %5.1f  max %5.1f | run time: median %5.1f  p90 %5.1f  max %5.1f | %d workgroups"
                  % (span, st[0], st[1], st[2], du[0], du[1], du[2], n))
