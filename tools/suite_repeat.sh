#!/bin/bash
# tools/suite_repeat.sh N TAG: the full `-m gpu` suite N times in a row on this box (no -x: one red test must not hide the ones
# behind it); per-run summary lines -> gpurun_out/<TAG>_suite_repeat.txt (copy to profiles/ to keep).
N=${1:-10}; TAG=${2:-r04}
mkdir -p gpurun_out
out=gpurun_out/${TAG}_suite_repeat.txt
{ echo "box: $(hostname)  $(date -u +%FT%TZ)  gpu unique id: $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | tr "\n" " ")  HEAD: $(cat .git/HEAD 2>/dev/null)"; } > $out
for i in $(seq 1 $N); do
  python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_suite_run_$i.log 2>&1
  echo "run $i: rc=$? $(tail -1 gpurun_out/${TAG}_suite_run_$i.log)" >> $out
  grep -E "^(FAILED|ERROR)" gpurun_out/${TAG}_suite_run_$i.log >> $out
done
cat $out
