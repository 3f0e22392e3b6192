#!/usr/bin/env python
"""Runs a few C3 frames (and optionally fwd+bwd iterations) for profiling under rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np
import torch
from bench import build_scene
from gaussianmesh_amd import scenes, rasterizer as Rz
from gaussianmesh_amd.deform import deform_tensors, sh_colors

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--bwd", type=int, default=0)
ap.add_argument("--gaussians", type=int, default=1_000_000)
a = ap.parse_args()
dev = torch.device("cuda:0")
P, W, H = a.gaussians, 1920, 1080
h = build_scene(P, W, H, 8)
g = {k: torch.tensor(v, device=dev) for k, v in h.items()}
ws = Rz.RasterWorkspace()
for i in range(a.frames):
    ms = g["mesh"][i % 8]
    dV = ms[:, 0:3].contiguous() - g["verts"]
    pos, cov, rot, cov6 = deform_tensors(g["tri"], g["weights"], dV, ms[:, 3:12].contiguous(), ms[:, 12:21].contiguous(), g["cov"], g["pos"])
    c = scenes.orbit_camera(i, 64, W, H)
    ct = {k: torch.tensor(c[k], device=dev) for k in ("view", "proj", "campos")}
    rgb = sh_colors(pos, ct["campos"], g["shs"], rot=rot, deg=3)
    Rz.rasterize_forward(torch.ones(3, device=dev), pos, rgb, g["opac"], None, None, 1.0, cov6, ct["view"], ct["proj"], c["tanx"], c["tany"],
                         H, W, None, 3, ct["campos"], False, False, workspace=ws)
if a.bwd:
    from gaussianmesh_amd import GaussianRasterizer, GaussianRasterizationSettings
    c = scenes.orbit_camera(0, 64, W, H)
    ct = {k: torch.tensor(c[k], device=dev) for k in ("view", "proj", "campos")}
    rs = GaussianRasterizationSettings(H, W, c["tanx"], c["tany"], torch.zeros(3, device=dev), 1.0, ct["view"], ct["proj"], 3, ct["campos"], False, False)
    leaves = [g[k].clone().requires_grad_(True) for k in ("pos", "opac", "shs", "scales", "rots")]
    m2d = torch.zeros_like(leaves[0], requires_grad=True)
    wgt = torch.randn((3, H, W), device=dev)
    for _ in range(a.bwd):
        color, _ = GaussianRasterizer(rs)(leaves[0], m2d, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
        (color * wgt).sum().backward()
torch.cuda.synchronize()
