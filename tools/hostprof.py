"""Host-side profile of bench.py's frame loop: a 2 k-Gaussian 128x128 scene keeps the GPU far from being the limit."""
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--gaussians", "2000", "--width", "128", "--height", "128", "--steps", "3000", "--warmup", "50", "--no-cpu-baseline", "--no-fwd-bwd", "--repeats", "0"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
