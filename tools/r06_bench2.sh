#!/bin/bash
# round 6: the driver's bench command twice on one box (spread of the line's numbers inside one box); tag = first argument
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_b2}
for r in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_${r}.json 2> gpurun_out/${tag}_bench_${r}.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_${r}.json"))
fb = d["fwd_bwd"]
print("run ${r}: value %.0f repeats %s | single %s | fwd_bwd %.3f train %.3f | c5 %.3f fixed %.3f phases %s" % (d["value"], d["repeats"]["frames_per_s"],
      {k: round(v, 4) for k, v in d["single_stream"].items() if isinstance(v, float)}, fb["ms_per_iter"], fb["ms_per_training_iteration"],
      d["c5"]["ms_per_iter"], d["c5_fixed"]["ms_per_iter"], d.get("c5_phases", {}).get("ms_per_iter")))
PY
done
