#!/bin/bash
# backward-blend timing of the library as built, no correctness check (for timing experiments whose results are wrong on purpose)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 20 --warmup 5 --repeats 0 --no-cpu-baseline --no-c5 --no-variants > gpurun_out/bt_$1.json 2> gpurun_out/bt_$1.err
python - <<PY
import json
f=json.load(open("gpurun_out/bt_$1.json"))["fwd_bwd"]
print("$1 fwd_bwd %.4f render_bwd %.4f render %.4f" % (f["ms_per_iter"], f["stage_ms"]["render_bwd"], f["stage_ms"]["render"]))
PY
