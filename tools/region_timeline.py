"""python tools/region_timeline.py [regions] [steps] [warmup]: per-frame completion times inside consecutive timed regions of bench.py's
pipelined loop (events recorded behind every frame on its stream) - is the FIRST region after the warm-up slower than the later ones,
and if so where in the region?  Patches bench.main's closures through the GM_BENCH_TIMELINE hook."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
regions = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
os.environ["GM_BENCH_TIMELINE"] = str(regions)
sys.argv = ["bench.py", "--steps", str(steps), "--warmup", str(warm), "--repeats", str(regions - 1), "--no-cpu-baseline", "--no-fwd-bwd", "--no-c5"]
import bench
bench.main()
