#!/bin/bash
# frames/s of the 100-step forward loop + the forward / backward blend times of the training iteration, for the library as built
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-c5 > gpurun_out/fb_$1.json 2> gpurun_out/fb_$1.err || tail -3 gpurun_out/fb_$1.err
python - <<PY
import json
d=json.load(open("gpurun_out/fb_$1.json")); f=d["fwd_bwd"]
print("$1", "fps %.0f" % d["value"], " ".join("%.0f" % x for x in d["repeats"]["frames_per_s"]), "| blend %.4f" % d["stage_ms"]["render"], "latency %.4f" % d["single_stream"]["ms_per_frame"],
      "| fwd_bwd %.4f" % f["ms_per_iter"], "train-fwd blend %.4f" % f["stage_ms"]["render"], "bwd blend %.4f" % f["stage_ms"]["render_bwd"], "c2 %.4f" % f["c2_500k_ms_per_iter"], "train %.4f" % f["ms_per_training_iteration"])
PY
