#!/bin/bash
# round 6: the driver's 20-step region at K = 4 / 5 / 7 / 8 frames per launch chain (is a burst of 20 frames better served by larger batches?)
cd $GRAFT_REPO_ROOT
run() {
  line=$(timeout 600 python bench.py --steps 20 --warmup 5 --repeats 4 --frames-per-launch $1 --streams $2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=%d streams=%d steps=%d value %.1f repeats %s discarded %s' % (d['config']['frames_per_launch'], d['config']['hip_streams'], d['steps'], d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['discarded_region_frames_per_s']))")
  echo "$line"
}
run 4 4; run 8 4; run 7 4; run 4 4; run 8 4; run 7 4; run 5 4; run 8 3
