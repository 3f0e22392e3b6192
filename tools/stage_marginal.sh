#!/bin/bash
# What each stage costs the PIPELINED loop: the library stops launching after a stage (GM_DEBUG_STOP_AFTER, gm_api.hip) and the
# 300-step loop is timed; differences between consecutive lines are the stages' marginal cost per frame in the 4-stream loop.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for stop in deform depth dup tile full; do
  if [ $stop = full ]; then unset GM_DEBUG_STOP_AFTER; else export GM_DEBUG_STOP_AFTER=$stop; fi
  python bench.py --steps 300 --warmup 20 --repeats 2 --no-cpu-baseline --no-fwd-bwd --no-c5 ${BENCH_ARGS} > gpurun_out/sm_$stop.json 2> gpurun_out/sm_$stop.err || tail -3 gpurun_out/sm_$stop.err
  python - <<PY
import json
d=json.load(open("gpurun_out/sm_$stop.json"))
f=[d["value"]]+d["repeats"]["frames_per_s"]
print("%-7s" % "$stop", "ms/frame " + " ".join("%.4f" % (1e3/x) for x in f), "| 1-stream %.4f" % d["single_stream"]["ms_per_frame"])
PY
done
done
