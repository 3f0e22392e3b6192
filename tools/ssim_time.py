#!/usr/bin/env python
"""ms per forward / backward launch of the photometric loss kernels (gm_ssim_fwd / gm_ssim_bwd) at 4K and 1080p."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianmesh_amd.loss import photometric_loss
for (W, H) in ((3840, 2160), (1920, 1080)):
    img = torch.rand((3, H, W), device="cuda", requires_grad=True); gt = torch.rand((3, H, W), device="cuda")
    def it():
        img.grad = None
        photometric_loss(img, gt, 0.2).backward()
    for _ in range(5): it()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 300
    e0.record()
    for _ in range(n): it()
    e1.record()
    torch.cuda.synchronize()
    print("%dx%d: %.4f ms per forward+backward of the loss (host clock), %.4f (device events)" % (W, H, 1e3 * (time.perf_counter() - t) / n, e0.elapsed_time(e1) / n))
