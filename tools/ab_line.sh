#!/bin/bash
# one line per run for tools/ab_flags.sh: pipelined frames/s and the single-stream stage times of `python bench.py $AB_BENCH_ARGS`
python bench.py --no-cpu-baseline --no-fwd-bwd --no-c5 $AB_BENCH_ARGS 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.readline())
print("fps %.1f latency_ms %.4f frac %.4f refused %s redone %s stage_ms %s" % (d["value"], d.get("single_stream",{}).get("ms_per_frame",0), d["frame_roofline"]["frac"], d["config"].get("frames_refused_by_direct_placement"), d["config"]["frames_redone"], json.dumps(d["stage_ms"])))'
