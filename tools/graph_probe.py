#!/usr/bin/env python
"""Experiment: what would capturing a whole frame (mesh_rs + fused deform/preprocess + ordering + blend, 12 launches) into ONE
HIP graph buy?  Per stream a torch.cuda.CUDAGraph over the library's own enqueue calls (they are capturable: no host
synchronisation in the sync-free path); per frame the camera and the deformed vertex positions are copied into the graph's fixed
input tensors, then the graph is replayed.  Reports single-stream latency and 4-stream throughput with and without graphs."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gaussianmesh_amd import rasterizer as Rz, scenes  # noqa: E402
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
tanx, tany = cams[0]["tanx"], cams[0]["tany"]
bg = torch.ones(3, device=dev)
hint = Rz.new_work_hint(W, H, dev)


def frame(ws, v1t, view, proj, campos):
    packed = mesh_rs_packed(g["verts"], v1t, g["faces"], adjacency)
    h = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], view, proj, tanx, tany, H, W, 3,
                                  campos, False, workspace=ws, want_count=False)
    out = h.finish(sync_free=True, image_only=True, work_hint=hint)
    ws.release(h)
    return out[1]


class Slot:
    def __init__(self):
        self.stream = torch.cuda.Stream(device=dev)
        self.ws = Rz.RasterWorkspace(growth=1.5)
        self.v1 = torch.empty_like(v1[0]); self.view = torch.empty_like(ct[0]["view"]); self.proj = torch.empty_like(ct[0]["proj"])
        self.campos = torch.empty_like(ct[0]["campos"])
        self.graph = None

    def load(self, i):
        self.v1.copy_(v1[i % F], non_blocking=True); self.view.copy_(ct[i % F]["view"], non_blocking=True)
        self.proj.copy_(ct[i % F]["proj"], non_blocking=True); self.campos.copy_(ct[i % F]["campos"], non_blocking=True)


def run(nstreams, use_graph, nframes=400):
    slots = [Slot() for _ in range(nstreams)]
    for s in slots:                                  # size the workspaces (exact path), then learn the capacity
        with torch.cuda.stream(s.stream):
            for i in range(F):
                h = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], mesh_rs_packed(g["verts"], v1[i], g["faces"], adjacency), g["cov"], g["pos"],
                                              g["shs"], g["opac"], ct[i]["view"], ct[i]["proj"], tanx, tany, H, W, 3, ct[i]["campos"], False, workspace=s.ws)
                h.finish(image_only=True, work_hint=hint)
    torch.cuda.synchronize()
    if use_graph:
        for s in slots:
            s.load(0)
            torch.cuda.synchronize()
            s.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(s.graph, stream=s.stream):
                s.out = frame(s.ws, s.v1, s.view, s.proj, s.campos)
        torch.cuda.synchronize()

    def issue(i):
        s = slots[i % nstreams]
        with torch.cuda.stream(s.stream):
            s.load(i)
            if use_graph:
                s.graph.replay()
            else:
                s.out = frame(s.ws, s.v1, s.view, s.proj, s.campos)
    for i in range(40):
        issue(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(nframes):
        issue(40 + i)
        if nstreams == 1:
            slots[0].stream.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return 1e3 * dt / nframes, slots[0].out.float().mean().item()


for ns in (1, 4):
    for ug in (False, True):
        ms, chk = run(ns, ug)
        print("streams %d graph %-5s: %.4f ms/frame (%.0f frames/s)  check %.6f" % (ns, ug, ms, 1e3 / ms, chk))
