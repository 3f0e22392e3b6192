#!/bin/bash
# round 6 experiment: the chip partitioned by CU masks (GM_EXP_CU_SPLIT = CUs of the ordering partition; read by a build of the library that
# existed for this measurement only: two masked streams + four events per batch inside gm_forward_deformed_batch_async): fused pass + blend of every batch on a stream that owns the other CUs.  Correctness of the stream hops first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_cu_split.txt
: > $out
GM_EXP_CU_SPLIT=64 timeout 600 python -m pytest tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -2 | tee -a $out
run() {
  line=$(env $1 timeout 600 python bench.py --steps 300 --warmup 20 --repeats 2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>gpurun_out/r06_cu_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$1] value %.1f repeats %s discarded %s' % (d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['discarded_region_frames_per_s']))")
  echo "$line" | tee -a $out
  grep -i -E "error|Traceback" gpurun_out/r06_cu_err.txt | head -3
}
run GM_X=0
run GM_EXP_CU_SPLIT=64
run GM_EXP_CU_SPLIT=32
run GM_EXP_CU_SPLIT=96
run "GM_EXP_CU_SPLIT=64 GM_EXP_CU_SPLIT_HIGH=1"
run GM_X=0
run GM_EXP_CU_SPLIT=48
run GM_EXP_CU_SPLIT=128
