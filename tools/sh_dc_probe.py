#!/usr/bin/env python
"""C5 at SH degree 0: what the SH rows cost as [N,16,3] (coefficient 0 of every 192-byte row: whole sectors) against a dense [N,1,3]
tensor (M = 1 through the operator's generic path) - preprocess forward / backward stage times and the Adam step.
    python tools/sh_dc_probe.py [N]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gaussianmesh_amd
gaussianmesh_amd.configure_runtime()
import numpy as np, torch
from gaussianmesh_amd import GaussianRasterizer, GaussianRasterizationSettings, scenes, _lib
from gaussianmesh_amd.model_ops import FusedAdam

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
W, H = 3840, 2160
dev = torch.device("cuda:0")
lib = _lib.lib()
sc = scenes.make_cloud(N, seed=0)
cam = scenes.orbit_camera(3, 64, W, H)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
rs = GaussianRasterizationSettings(H, W, cam["tanx"], cam["tany"], torch.zeros(3, device=dev), 1.0, t(cam["view"]), t(cam["proj"]), 0, t(cam["campos"]), False, False, None)
rast = GaussianRasterizer(rs)
base = [t(sc[k]).requires_grad_(True) for k in ("means", "opac", "scales", "rots")]
m2 = torch.zeros_like(base[0], requires_grad=True)
wgt = torch.randn((3, H, W), device=dev)
full = t(sc["shs"])


def run(shs, tag):
    shs = shs.clone().requires_grad_(True)
    def it():
        for l in base + [m2, shs]:
            l.grad = None
        color, _ = rast(base[0], m2, base[1], shs=shs, scales=base[2], rotations=base[3])
        (color * wgt).sum().backward()
    for _ in range(5):
        it()
    torch.cuda.synchronize()
    lib.gm_profile_reset(); lib.gm_profile_enable(1)
    n = 20
    for _ in range(n):
        it()
    torch.cuda.synchronize()
    lib.gm_profile_enable(0)
    out = {}
    for st in ("preprocess", "preprocess_bwd"):
        ms = C.c_double(0); k = C.c_int64(0)
        lib.gm_profile_read(st.encode(), C.byref(ms), C.byref(k))
        out[st] = ms.value / n
    # Adam on this tensor alone
    g = {"params": [torch.nn.Parameter(shs.detach().clone())], "lr": 2.5e-3, "name": "f"}
    if shs.shape[1] == 16:
        g.update(lr_rest=1.25e-4, period=48, split=3, active=3)
    g["params"][0].grad = torch.randn_like(g["params"][0]) * 1e-3
    opt = FusedAdam([g], eps=1e-15)
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        opt.step()
    e1.record(); torch.cuda.synchronize()
    out["adam_sh"] = e0.elapsed_time(e1) / 50
    print(tag, " ".join("%s %.4f ms" % kv for kv in out.items()), "grad", float(shs.grad.abs().sum()))


run(full, "rows [N,16,3], degree 0:")
run(full[:, :1].contiguous(), "dense [N,1,3], degree 0: ")
