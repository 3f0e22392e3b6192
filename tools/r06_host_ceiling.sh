#!/bin/bash
# round 6: is the batched loop bound by the HOST?  The same loop on a cloud whose device work is negligible runs at the host's pace
# (launch chain per batch, status reads, Python): frames/s there = the ceiling the host sets for any device-side improvement.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_host_ceiling.txt
: > $out
run() {
  line=$(timeout 600 python bench.py --steps 300 --warmup 20 --repeats 2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants "$@" 2>gpurun_out/r06_host_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('P=%d %dx%d K=%d streams=%d: %.1f frames/s = %.1f us per frame, repeats %s' % (d['config']['gaussians'], d['config']['width'], d['config']['height'], d['config']['frames_per_launch'], d['config']['hip_streams'], d['value'], 1e6/d['value'], d.get('repeats',{}).get('frames_per_s')))")
  echo "$line" | tee -a $out
  grep -i -E "error|Traceback" gpurun_out/r06_host_err.txt | head -3
}
run --gaussians 20000 --width 320 --height 200
run --gaussians 20000 --width 320 --height 200 --frames-per-launch 8
run --gaussians 20000 --width 320 --height 200 --frames-per-launch 1
run
run --frames-per-launch 8
