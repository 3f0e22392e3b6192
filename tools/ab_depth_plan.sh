# usage (GPU box, repo root): tools/ab_depth_plan.sh <tag> [reps]; direct depth placement (DepthPlan) against the depth partition,
# interleaved runs of the default pipelined loop and the single-stream stage times -> gpurun_out/<tag>_ab_depth_plan.txt
tag=${1:-r04}; reps=${2:-3}
out=gpurun_out/${tag}_ab_depth_plan.txt
mkdir -p gpurun_out; : > $out
pick='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], "fps %.1f" % d["value"], "latency_ms %.4f" % d.get("single_stream", {}).get("ms_per_frame", 0), "frac %.4f" % d["frame_roofline"]["frac"], "refused", d["config"].get("frames_refused_by_direct_placement"), "redone", d["config"]["frames_redone"], "stage_ms", json.dumps(d["stage_ms"]))'
for r in $(seq $reps); do
  python bench.py --no-cpu-baseline --no-fwd-bwd --no-c5 --depth-plan 2>/dev/null | python -c "$pick" direct >> $out
  python bench.py --no-cpu-baseline --no-fwd-bwd --no-c5 2>/dev/null | python -c "$pick" partition >> $out
done
python bench.py --no-cpu-baseline --no-fwd-bwd --no-c5 --streams 1 --exact-count --depth-plan 2>/dev/null | python -c "$pick" direct_1stream >> $out
python bench.py --no-cpu-baseline --no-fwd-bwd --no-c5 --streams 1 --exact-count 2>/dev/null | python -c "$pick" partition_1stream >> $out
cat $out
