#!/usr/bin/env python
"""Splat-size statistics of one C5 frame: how many Gaussians take the large-rectangle path of the instance emission."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianmesh_amd import rasterizer as Rz, scenes  # noqa: E402

dev = torch.device("cuda:0")
W, H = 3840, 2160
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
b = scenes.make_cloud(1_000_000, seed=1, extent=1.0)
nb = np.linalg.norm(b["means"], axis=1, keepdims=True) + 1e-6
means = b["means"] / nb * (6 + 6 * nb)
cam = scenes.orbit_camera(3, 32, W, H)
ct = {n: t(cam[n]) for n in ("view", "proj", "campos")}
for pol in (2, 3):
    nr, color, radii, *_ = Rz.rasterize_forward(torch.zeros(3, device=dev), t(means), None, t(b["opac"]), t(b["scales"]), t(b["rots"]), 1.0, None,
                                                ct["view"], ct["proj"], cam["tanx"], cam["tany"], H, W, t(b["shs"]), 3, ct["campos"], False, False,
                                                emission_policy=pol)
    r = radii.cpu().numpy()
    print("policy", pol, "background cloud alone: visible", (r > 0).sum(), "instances", nr, "radius > 56 px:", (r > 56).sum(), "> 200:", (r > 200).sum(),
          "> 1000:", (r > 1000).sum(), "max", r.max())
