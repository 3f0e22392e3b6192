#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_k}
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_model_ops.py tests/test_gpu_bg_route.py tests/test_gpu_topology.py -q -x > gpurun_out/${tag}_train_tests.txt 2>&1; tail -15 gpurun_out/${tag}_train_tests.txt
python - <<PY
import json, time, sys
sys.path.insert(0, ".")
import bench, torch
dev = torch.device("cuda", 0)
r = {}
r["c5_fixed"] = bench.c5_leg(100, 10, dev=dev)["ms_per_iter"]
r["c5_phases"] = bench.c5_phases(100, 20, dev=dev)
r["c5_phases_unfused"] = bench.c5_phases(100, 20, dev=dev, degrees=(3,), trainer_kw=dict(fused_sh_step=False))
print(json.dumps(r))
json.dump(r, open("gpurun_out/${tag}_c5.json", "w"))
PY
