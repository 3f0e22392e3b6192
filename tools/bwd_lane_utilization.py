#!/usr/bin/env python
"""How full are the backward blend's waves?  CPU only (oracle forward of the C3 bench frame, then numpy over a sample of tiles): for every 8x8
quadrant the list prefix up to its deepest contributor is walked as render_bwd_kernel walks it; an entry is TAKEN when at least one pixel of
the quadrant accepts it (alpha >= 1/255, power <= 0, position below the pixel's last contributor).  Prints the share of visited entries that
are taken and how many of the 64 pixels take a taken entry - the lane utilisation of phase 1 (lane = pixel).
Round 6, 1 M Gaussians / 1080p, 400 sampled tiles: 26 % of the visited entries are taken (the quadrant bits and may_touch remove most of the
rest before the walk), a taken entry is taken by 30.2 of 64 pixels on average (47 %); 26 % of them by at most 8 pixels, 17 % by all 64."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gaussianmesh_amd import scenes
from oracle import oracle

P, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1920, 1080
host = bench.build_scene(P, W, H, 4)
cam = scenes.orbit_camera(1, 64, W, H)
sc = dict(means=host["pos"], opac=host["opac"], shs=host["shs"],
          cov3D_precomp=scenes.strip_symmetric(host["cov"].reshape(P, 3, 3).astype(np.float64)).astype(np.float32))
fw = oracle.forward_full(sc, cam, np.zeros(3, np.float32), D=3, use_precomp_cov=True)
geo, bins = fw["geo"], fw["bins"]
nc = fw["n_contrib"].reshape(H, W)
gx = (W + 15) // 16
xy = geo["xy"].astype(np.float64); co = geo["conic_op"].astype(np.float64)
lens = bins["ranges"][:, 1] - bins["ranges"][:, 0]
sample = np.random.default_rng(0).choice(np.nonzero(lens > 0)[0], size=400, replace=False)
visited = taken = lanes = 0
hist = np.zeros(65, np.int64)
for t in sample:
    ty, tx = divmod(int(t), gx)
    r0, r1 = bins["ranges"][t]
    g = bins["point_list"][r0:r1]
    for q in range(4):
        x0, y0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        if x0 >= W or y0 >= H:
            continue
        px, py = np.meshgrid(np.arange(x0, min(x0 + 8, W)), np.arange(y0, min(y0 + 8, H)))
        px, py = px.reshape(-1), py.reshape(-1)
        last = nc[py, px].astype(np.int64)
        start = int(last.max())
        if start == 0:
            continue
        gg = g[:start]
        dx = xy[gg, 0][:, None] - px[None, :]; dy = xy[gg, 1][:, None] - py[None, :]
        power = -0.5 * (co[gg, 0][:, None] * dx * dx + co[gg, 2][:, None] * dy * dy) - co[gg, 1][:, None] * dx * dy
        alpha = np.minimum(0.99, co[gg, 3][:, None] * np.exp(np.minimum(power, 0)))
        cnt = ((power <= 0) & (alpha >= 1 / 255) & (np.arange(start)[:, None] < last[None, :])).sum(1)
        visited += start; taken += int((cnt > 0).sum()); lanes += int(cnt.sum())
        hist += np.bincount(cnt, minlength=65)[:65]
c = np.cumsum(hist[1:]) / hist[1:].sum()
print("%d Gaussians, %dx%d, %d tiles sampled: %d quadrant-entries visited, %.1f %% taken; %.1f of 64 pixels take a taken entry (%.1f %%); "
      "taken by <= 8 pixels: %.0f %%, <= 16: %.0f %%, <= 32: %.0f %%, all 64: %.0f %%" % (P, W, H, len(sample), visited, 100 * taken / visited, lanes / taken,
                                                                                   100 * lanes / taken / 64, 100 * c[7], 100 * c[15], 100 * c[31], 100 * hist[64] / hist[1:].sum()))
