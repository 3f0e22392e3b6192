#!/bin/bash
# timing of the backward blend for the library as built (bench fwd_bwd legs only)
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-x}
python -m pytest tests/test_gpu_parity.py -x -q -k "backward or grad" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --repeats 0 --no-cpu-baseline --no-c5 > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/ab_$tag.json"))
f=d["fwd_bwd"]
print("$tag", "fwd_bwd %.4f" % f["ms_per_iter"], "render_bwd %.4f" % f["stage_ms"]["render_bwd"], "c2 %.4f" % f["c2_500k_ms_per_iter"], "train %.4f" % f["ms_per_training_iteration"], "fps %.0f" % d["value"])
PY
