#!/usr/bin/env python
"""Per-wave timeline of one render_fwd_kernel launch on a C3 frame (gm_debug_render_trace): when waves start and end, how the
kernel's duration splits into the bulk and the tail, and which list lengths the late waves have."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gaussianmesh_amd import _lib, rasterizer as Rz, scenes  # noqa: E402
from gaussianmesh_amd.deform import pack_mesh_state  # noqa: E402


def main():
    P, W, H, F = 1_000_000, 1920, 1080, 64
    dev = torch.device("cuda:0")
    host = bench.build_scene(P, W, H, F)
    g = {k: torch.tensor(host[k], device=dev) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = torch.tensor(host["tri"], dtype=torch.int32, device=dev)
    lib = _lib.lib()
    lib._handle  # noqa
    fn = lib.gm_debug_render_trace
    fn.restype = None; fn.argtypes = [C.c_void_p]
    nblocks = 4 * 2048 * 4 + 64
    buf = torch.zeros((nblocks * 4 * 8,), dtype=torch.int64, device=dev)
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cam = scenes.orbit_camera(k, F, W, H)
    ct = {n: torch.tensor(cam[n], device=dev) for n in ("view", "proj", "campos")}
    packed = pack_mesh_state(torch.tensor(host["mesh"][k], device=dev), g["verts"])
    bg = torch.ones(3, device=dev)
    # argv[2] = "hint": dispatch order from a work hint filled by the two frames before the traced one (same camera: the best
    # the hint can be), otherwise by list length
    hint = Rz.new_work_hint(W, H, dev) if len(sys.argv) > 2 and sys.argv[2] == "hint" else None
    for rep in range(3):
        if rep == 2:
            buf.zero_(); fn(buf.data_ptr())
        Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], ct["view"], ct["proj"],
                                  cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish(work_hint=hint)
        torch.cuda.synchronize()
    fn(None)
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0; end = (t[:, 1] - t0) / 100.0; n = t[:, 2] & 0xFFFFFF        # wall_clock64 ticks at 100 MHz -> microseconds
    useful = (t[:, 2] >> 24) & 0xFFFFF; lanes = t[:, 2] >> 44
    dur = end - start
    print("waves %d, kernel span %.1f us, sum of wave durations %.0f us (= %.1f us x 8192 resident slots)" % (len(t), end.max(), dur.sum(), dur.sum() / 8192))
    for q in (50, 90, 99, 99.9):
        print("  %5.1f %% of the waves have ended by %.1f us; wave duration percentile %.1f us" % (q, np.percentile(end, q), np.percentile(dur, q)))
    edges = np.arange(0, end.max() + 10, 10)
    alive = [(np.logical_and(start <= a, end > a)).sum() for a in edges]
    print("waves alive at t = 0, 10, 20 ... us:", alive)
    late = np.argsort(-end)[:12]
    print("latest waves: end us / duration us / list length:", [(round(end[i], 1), round(dur[i], 1), int(n[i])) for i in late])
    iters = t[:, 3] & 0xFFFF; cand = (t[:, 3] >> 16) & 0xFFFFFF; surv = t[:, 3] >> 40
    print("totals: iterations %d, candidates %d, survivors %d, survivors some pixel accepts %d, accepting lanes per such survivor %.1f" %
          (iters.sum(), cand.sum(), surv.sum(), useful.sum(), lanes.sum() / max(useful.sum(), 1)))
    tb = t[:, 4] & 0xFFFFFFFF; lr = t[:, 4] >> 32; q4 = t[:, 5] & 0xFFFFFFFF; steps = t[:, 5] >> 32
    print("survivor-loop steps (4 survivors each) %d; ideal list entries to walk if the wave's pixels were split top/bottom %d, left/right %d, "
          "in four 4x4 blocks %d (max over the parts per batch, entries some pixel of the part accepts)" % (steps.sum(), tb.sum(), lr.sum(), q4.sum()))
    print("latest waves: iterations / candidates / survivors:", [(int(iters[i]), int(cand[i]), int(surv[i])) for i in late])
    for lo, hi in ((0, 1), (1, 16), (16, 64), (64, 128), (128, 256), (256, 512), (512, 4096)):
        m = (surv >= lo) & (surv < hi)
        if m.any():
            print("  survivors [%4d, %4d): %6d waves, mean duration %6.1f us, mean iterations %5.1f, us per survivor %.3f" %
                  (lo, hi, m.sum(), dur[m].mean(), iters[m].mean(), dur[m].sum() / max(surv[m].sum(), 1)))
    longl = n > np.percentile(n, 99)
    print("lists above the 99th percentile length (%d entries): mean duration %.1f us, max %.1f us" % (np.percentile(n, 99), dur[longl].mean(), dur[longl].max()))


if __name__ == "__main__":
    main()
