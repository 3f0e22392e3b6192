#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_w}
out=gpurun_out/${tag}_batch_ab.txt
: > $out
run() {
  line=$(timeout 600 python bench.py --steps $3 --warmup $4 --repeats 3 --frames-per-launch $1 --streams $2 $5 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>gpurun_out/${tag}_ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=%d streams=%d steps=%d $5 value %.1f repeats %s discarded %s' % (d['config']['frames_per_launch'], d['config']['hip_streams'], d['steps'], d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['discarded_region_frames_per_s']))")
  echo "$line" | tee -a $out
  grep -i -E "error|Traceback" gpurun_out/${tag}_ab_err.txt | head -3
}
run 4 4 20 5 ""
run 4 4 20 5 "--ramp"
run 4 4 20 5 ""
run 4 4 20 5 "--ramp"
run 6 4 20 5 "--ramp"
run 4 4 300 20 "--ramp"
