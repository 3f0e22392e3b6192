#!/bin/bash
# round 6: the batch tests, then the pipelined loop at frames-per-launch 1 / 2 / 4 over 1 .. 4 streams (300-step regions, repeats)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_b}
timeout 900 python -m pytest tests/test_gpu_batch.py -q -x -s > gpurun_out/${tag}_batch_tests.txt 2>&1
tail -15 gpurun_out/${tag}_batch_tests.txt
out=gpurun_out/${tag}_batch_ab.txt
: > $out
for cfg in "1 4" "4 1" "4 2" "4 3" "4 4" "2 2" "2 4" "3 3" "1 4" "4 2"; do
  set -- $cfg
  line=$(timeout 600 python bench.py --steps 300 --warmup 20 --repeats 2 --frames-per-launch $1 --streams $2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>gpurun_out/${tag}_ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=%d streams=%d value %.1f repeats %s redone %s single %.4f' % (d['config']['frames_per_launch'], d['config']['hip_streams'], d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['frames_redone'], d['single_stream']['ms_per_frame']))")
  echo "$line" | tee -a $out
  tail -3 gpurun_out/${tag}_ab_err.txt | grep -i -E "error|Traceback" 
done
