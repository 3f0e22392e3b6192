#!/bin/bash
# A/B of the backward blend on the GPU box: parity tests, then fwd_bwd timings of bench.py with and without GM_BWD_V1
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py -x -q -k "backward or grad or culling" 2>&1 | tail -5
for v in new v1; do
  if [ $v = v1 ]; then export GM_BWD_V1=1; else unset GM_BWD_V1; fi
  python bench.py --steps 20 --warmup 5 --repeats 0 --no-cpu-baseline --no-c5 > gpurun_out/ab_bwd_$v.json 2> gpurun_out/ab_bwd_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_bwd_$v.json"))
f=d["fwd_bwd"]
print("$v", "fwd_bwd %.4f" % f["ms_per_iter"], "render_bwd %.4f" % f["stage_ms"]["render_bwd"], "c2 %.4f" % f["c2_500k_ms_per_iter"], "train %.4f" % f["ms_per_training_iteration"])
PY
done
