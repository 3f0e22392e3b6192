"""Host-side profile of Trainer.step: a 20 k-Gaussian 320x200 scene keeps the GPU far from being the limit, so the loop's rate IS the host's
time per iteration (python + autograd + ctypes + launches).  Prints us per iteration and the top functions by own time."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

tr, cams, target, zero = bench.build_c5(20000, 6000, 320, 200, torch.device("cuda", 0), True, ncams=8)
for i in range(50):
    tr.step(cams[i % 8], target, zero)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t = time.perf_counter()
for i in range(n):
    tr.step(cams[i % 8], target, zero)
torch.cuda.synchronize()
print("host-bound iteration: %.1f us (sh steps inside the backward: %d of %d)" % (1e6 * (time.perf_counter() - t) / n, tr.sh_steps_fused, tr.optimizer.n_step))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    tr.step(cams[i % 8], target, zero)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
