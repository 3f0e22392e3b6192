#!/bin/bash
# round 6: stable ranks of the scatter passes from LDS atomics + one ballot per group (GM_SCATTER_RANK=1) against match-any over the digit
# bits (0): list tests first, then the loop A/B (tools/ab_build.sh) and the per-kernel durations / instruction counts of both builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_sc}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_direct_depth.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/${tag}_tests.txt 2>&1
tail -3 gpurun_out/${tag}_tests.txt
bash tools/ab_build.sh "-DGM_SCATTER_RANK=0" "-DGM_SCATTER_RANK=1" > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
