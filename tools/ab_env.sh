#!/bin/bash
# A/B of an environment knob on the pipelined forward loop in ONE gpurun call: tools/ab_env.sh VAR v1 v2 ... ("-" = unset); two interleaved rounds
cd ${GRAFT_REPO_ROOT:-.}
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    python bench.py --steps 300 --warmup 20 --repeats 2 --no-cpu-baseline --no-fwd-bwd --no-c5 ${BENCH_ARGS} > gpurun_out/abe.json 2> gpurun_out/abe.err || tail -3 gpurun_out/abe.err
    python - <<PY
import json
d=json.load(open("gpurun_out/abe.json"))
print("$var=$v", "%.0f" % d["value"], " ".join("%.0f" % x for x in d["repeats"]["frames_per_s"]), "1-stream %.4f" % d["single_stream"]["ms_per_frame"], "deform %.4f" % d["stage_ms"]["deform"])
PY
  done
done
