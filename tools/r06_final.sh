#!/bin/bash
# round 6, final: the driver's bench command timed, with its rocprof kernel table (and a one-stream table of the batched loop)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_final}
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_20steps.json 2> gpurun_out/${tag}_bench_20steps.err
echo "bench.py --steps 20 --warmup 5: $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_20steps.json"))
print("value", d["value"], "repeats", d.get("repeats"), "variants", d.get("variants"), "discarded", d["config"].get("discarded_region_frames_per_s"))
print("single", d.get("single_stream", {}).get("ms_per_frame"), "stage_ms", d.get("stage_ms"))
r = d["roofline"]; print("roofline", r["kernel_name"], r["frac"], r["avg_ms"], r["traffic"], r["algorithmic_bytes"], r.get("valu_issue", {}).get("frac_of_kernel_time"), "frame", d["frame_roofline"]["frac"], d["batch"]["frame_roofline_frac"])
fb = d["fwd_bwd"]; print("fwd_bwd", fb["ms_per_iter"], fb["stage_ms"], "loss", fb["ms_per_iter_with_l1_ssim_loss"], "train", fb["ms_per_training_iteration"], fb["ms_per_training_iteration_sync_free"], "c2", fb["c2_500k_ms_per_iter"])
print("c5", d["c5"]["ms_per_iter"], "r04 scene", d["c5_r04_scene"]["ms_per_iter"], "fixed", d["c5_fixed"]["ms_per_iter"], "phases", {k: v for k, v in d["c5_phases"].items() if k.startswith("deg") and not k.endswith(("redone", "operand"))})
print("cpu", d["cpu_baseline"])
PY
t0=$(date +%s)
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
echo "bench.py (defaults: 300 steps): $(( $(date +%s) - t0 )) s"
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_default.json')); print('300 steps: value', d['value'], d.get('repeats'), d.get('variants'), 'frame frac', d['frame_roofline']['frac'])"
PROF_LINES=70 bash tools/prof.sh ${tag}_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c5 > /dev/null 2>&1
PROF_LINES=70 bash tools/prof.sh ${tag}_k4_1stream --steps 100 --warmup 20 --repeats 0 --streams 1 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants > /dev/null 2>&1
tail -30 gpurun_out/${tag}_k4_1stream_kernel_stats.txt
