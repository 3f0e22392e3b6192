#!/bin/bash
# round 6 experiment: fewer one-wave workgroups of the blend / the fused pass per CU (extra dynamic LDS per workgroup, GM_EXP_*_LDS_PAD, read by a build of the library that existed for this measurement only),
# so that the ordering launches of the other streams find registers and wave slots - frames/s of the four-stream loop, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_pad}
out=gpurun_out/${tag}_lds_pad_ab.txt
: > $out
run() {   # blend pad, fused pad, steps
  line=$(GM_EXP_FWD_LDS_PAD=$1 GM_EXP_FUSED_LDS_PAD=$2 timeout 600 python bench.py --steps $3 --warmup 20 --repeats 2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>gpurun_out/${tag}_ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('blend pad $1 fused pad $2 steps=%d value %.1f repeats %s discarded %s single-stream %.4f ms' % (d['steps'], d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['discarded_region_frames_per_s'], d['single_stream']['ms_per_frame']))")
  echo "$line" | tee -a $out
  grep -i -E "error|Traceback" gpurun_out/${tag}_ab_err.txt | head -3
}
run 0 0 300
run 5056 0 300      # blend: 10 KiB per wave -> 16 waves per CU (4 per SIMD instead of 5)
run 8128 0 300      # 13 KiB -> 12 per CU (3 per SIMD)
run 0 4096 300      # fused pass: 16 KiB -> 10 per CU
run 5056 4096 300
run 0 0 300
run 2496 0 300      # 7.5 KiB -> 21 per CU (registers allow 20)
run 5056 0 300
