#!/usr/bin/env python
"""Per-tile list statistics of one C3 frame: list length, processed length (max n_contrib per 8x8 quadrant), survivors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import build_scene
from gaussianmesh_amd import scenes, rasterizer as Rz, _lib
from gaussianmesh_amd.deform import deform_shade

dev = torch.device("cuda:0")
P, W, H = 1_000_000, 1920, 1080
h = build_scene(P, W, H, 8)
g = {k: torch.tensor(v, device=dev) for k, v in h.items()}
lib = _lib.lib()
for frame in (0, 5):
    ms = g["mesh"][frame % 8]
    dV = ms[:, 0:3].contiguous() - g["verts"]
    c = scenes.orbit_camera(frame, 64, W, H)
    ct = {k: torch.tensor(c[k], device=dev) for k in ("view", "proj", "campos")}
    pos, cov6, rgb = deform_shade(g["tri"], g["weights"], dV, ms[:, 3:12].contiguous(), ms[:, 12:21].contiguous(), g["cov"], g["pos"], g["shs"], ct["campos"], deg=3)
    nr, color, radii, geom, binning, img = Rz.rasterize_forward(torch.ones(3, device=dev), pos, rgb, g["opac"], None, None, 1.0, cov6, ct["view"], ct["proj"],
                                                               c["tanx"], c["tany"], H, W, None, 3, ct["campos"], False, False)
    torch.cuda.synchronize()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gx * gy
    import ctypes
    def view(buf, ptr, n, dt):
        off = ptr - buf.data_ptr()
        return buf[off:off + n * 4].view(dt).cpu().numpy()
    rng = view(img, lib.gm_image_field(img.data_ptr(), W, H, b"ranges"), tiles * 2, torch.int32).reshape(tiles, 2)
    nc = view(img, lib.gm_image_field(img.data_ptr(), W, H, b"n_contrib"), W * H, torch.int32).reshape(H, W)
    fT = view(img, lib.gm_image_field(img.data_ptr(), W, H, b"final_T"), W * H, torch.float32).reshape(H, W)
    n = (rng[:, 1] - rng[:, 0]).astype(np.int64)
    Hp, Wp = gy * 16, gx * 16
    ncp = np.zeros((Hp, Wp), np.int64); ncp[:H, :W] = nc
    q = ncp.reshape(gy * 2, 8, gx * 2, 8).max(axis=(1, 3))          # per 8x8 quadrant: last contributing position
    sat = np.ones((Hp, Wp), bool); sat[:H, :W] = fT < 1e-4 * 1.0   # approx: saturated pixels
    print("frame", frame, "R", nr, "tiles", tiles)
    print(" list length: mean %.0f  p50 %d p90 %d p99 %d max %d" % (n.mean(), *np.percentile(n, [50, 90, 99]).astype(int), n.max()))
    print(" quadrant last-contributor pos: mean %.0f p50 %d p90 %d p99 %d max %d" % (q.mean(), *np.percentile(q, [50, 90, 99]).astype(int), q.max()))
    nq = np.repeat(np.repeat(n.reshape(gy, gx), 2, 0), 2, 1)
    print(" sum over quadrants of list length %.2fM, of last-contributor pos %.2fM" % (nq.sum() / 1e6, q.sum() / 1e6))
    Tp = np.zeros((Hp, Wp), np.float32); Tp[:H, :W] = fT
    tmax = Tp.reshape(gy * 2, 8, gx * 2, 8).max(axis=(1, 3))
    for thr in (1e-3, 2e-2, 0.2):
        satq = tmax < thr
        proc = np.where(satq, np.minimum(np.ceil(q / 64) + 1, np.ceil(nq / 64)), np.ceil(nq / 64))
        print("  thr %g: saturated quadrants %.1f%% (of non-empty %.1f%%); batches processed %.0fk (saturated part %.0fk, unsaturated %.0fk)" % (
            thr, 100 * satq.mean(), 100 * (satq & (nq > 0)).sum() / max((nq > 0).sum(), 1), proc.sum() / 1e3, proc[satq].sum() / 1e3, proc[~satq].sum() / 1e3))
    # unsaturated quadrants: how many of their 64 pixels stay live to the end of the list (they bound the walk)
    livecnt = (Tp.reshape(gy * 2, 8, gx * 2, 8) >= 1e-3).sum(axis=(1, 3))
    long_q = (~(tmax < 2e-2)) & (nq > 2000)
    if long_q.any():
        lc = livecnt[long_q]
        print("  unsaturated quadrants on lists > 2000: %d; live pixels at the end: mean %.1f  p50 %d  p90 %d  max %d;  <=16 live: %.0f%%  <=32 live: %.0f%%" % (
            long_q.sum(), lc.mean(), *np.percentile(lc, [50, 90]).astype(int), lc.max(), 100 * (lc <= 16).mean(), 100 * (lc <= 32).mean()))
    top = np.argsort(-n)[:5]
    print(" longest tiles:", [(int(t % gx), int(t // gx), int(n[t]), int(q.reshape(gy, 2, gx, 2)[t // gx, :, t % gx, :].max())) for t in top])
    print(" batches per wave: total %.0fk, max %d" % (np.ceil(nq / 64).sum() / 1e3, int(np.ceil(n.max() / 64))))
