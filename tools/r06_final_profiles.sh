#!/bin/bash
# round 6: (1) the driver's 20-step line and the 300-step loop (profiles/r06_batch_plan_ab.txt was this script's A/B of a batch planner that has since been removed), (2) HBM traffic of the batched loop per launch (grid z split), (3) rocprof kernel
# table of the driver's command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_p}
out=gpurun_out/${tag}_plan_ab.txt
: > $out
run() {
  line=$(timeout 600 python bench.py --steps $1 --warmup $2 --repeats 3 $3 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants 2>gpurun_out/${tag}_ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps=%d plan=%s value %.1f repeats %s discarded %s redone %s roofline %.3f %s' % (d['steps'], d['config']['frames_per_launch'], d['value'], d.get('repeats',{}).get('frames_per_s'), d['config']['discarded_region_frames_per_s'], d['config']['frames_redone'], d['roofline']['frac'], d['roofline']['kernel_name'][:60]))")
  echo "$line" | tee -a $out
  grep -i -E "error|Traceback" gpurun_out/${tag}_ab_err.txt | head -3
}
run 20 5 ""
run 300 20 ""
PMC_TIMEOUT=400 PMC_BENCH_ARGS="--no-fwd-bwd --no-c5 --no-variants --repeats 0" PMC_MODE_ARGS="--streams 1 --frames-per-launch 4" PMC_STEPS=8 PMC_FILTER="" tools/pmc.sh ${tag}_batch_hbm_traffic FETCH_SIZE WRITE_SIZE > /dev/null 2>&1
python tools/hbm_traffic.py gpurun_out/${tag}_batch_hbm_traffic_pmc.txt gpurun_out/${tag}_hbm_traffic_batch.json 1000000 1920 1080 batch4 | head -60
PROF_LINES=60 bash tools/prof.sh ${tag}_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-bwd --no-c5 | tail -45
