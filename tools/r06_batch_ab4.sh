#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_v}
out=gpurun_out/${tag}_batch_ab.txt
: > $out
run() {
  line=$(timeout 600 python bench.py --steps $3 --warmup $4 --repeats 2 --frames-per-launch $1 --streams $2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>gpurun_out/${tag}_ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=%d streams=%d steps=%d value %.1f repeats %s discarded %s redone %s' % (d['config']['frames_per_launch'], d['config']['hip_streams'], d['steps'], d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['discarded_region_frames_per_s'], d['config']['frames_redone']))")
  echo "$line" | tee -a $out
  grep -i -E "error|Traceback" gpurun_out/${tag}_ab_err.txt | head -3
}
run 4 4 300 20
run 5 4 300 20
run 6 4 300 20
run 8 4 300 20
run 8 2 300 20
run 6 3 300 20
run 4 4 20 5
run 5 4 20 5
run 8 2 20 5
run 5 4 20 5
run 4 4 20 5
