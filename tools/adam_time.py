#!/usr/bin/env python
"""Time of one FusedAdam step over the trainable tensors of a mesh-bound model (N Gaussians x 60 parameters), HIP events."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gaussianmesh_amd.model_ops import FusedAdam

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda:0")
shapes = [("bc", (N, 3)), ("distance", (N, 1)), ("scaling", (N, 3)), ("rotation", (N, 4)), ("opacity", (N, 1)), ("f_dc+f_rest", (N, 16, 3))]
groups = []
for name, sh in shapes:
    p = torch.nn.Parameter(torch.randn(sh, device=dev))
    p.grad = torch.randn(sh, device=dev) * 1e-3
    g = {"params": [p], "lr": 1e-3, "name": name}
    if name.startswith("f_dc"):
        g.update(lr_rest=5e-5, period=48, split=3)
    groups.append(g)
opt = FusedAdam(groups, eps=1e-15)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    opt.step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nparam = N * 60
print("N %d: %.4f ms per step, %.2f TB/s (28 B per parameter)" % (N, ms, nparam * 28 / ms / 1e9))
