#!/bin/bash
# round 6: the full -m gpu suite and the driver's bench line in one call (tag = first argument)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_a}
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/${tag}_suite.txt 2>&1
tail -6 gpurun_out/${tag}_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_20steps.json 2> gpurun_out/${tag}_bench_20steps.err
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_20steps.json"))
print("value", d["value"], "repeats", d.get("repeats"), "variants", d.get("variants"), "discarded", d["config"].get("discarded_region_frames_per_s"))
print("single", d.get("single_stream"), "stage_ms", d.get("stage_ms"))
print("roofline", d["roofline"]["frac"], "frame", d["frame_roofline"]["frac"])
fb = d["fwd_bwd"]; print("fwd_bwd", fb["ms_per_iter"], fb["stage_ms"], "loss", fb["ms_per_iter_with_l1_ssim_loss"], "train", fb["ms_per_training_iteration"], "c2", fb["c2_500k_ms_per_iter"])
print("c5", d["c5"]["ms_per_iter"], "r04 scene", d["c5_r04_scene"]["ms_per_iter"], "fixed", d["c5_fixed"]["ms_per_iter"])
print("cpu", d["cpu_baseline"])
PY
