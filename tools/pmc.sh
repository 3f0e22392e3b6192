#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
# rocprofv3 --pmc passes over a short single-stream bench run; per-kernel means -> gpurun_out/<tag>_pmc.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_pmc.txt
: > $out
i=0
for pass in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  timeout ${PMC_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_${tag}_$i -o p -- python ${PMC_SCRIPT:-$root/bench.py} --no-cpu-baseline ${PMC_BENCH_ARGS---no-fwd-bwd} ${PMC_MODE_ARGS---streams 1 --exact-count} --steps ${PMC_STEPS:-6} --warmup 2 > /dev/null 2> /tmp/pmc_${tag}_$i.err
  db=$(find /tmp/pmc_${tag}_$i -name "*.db" | head -1)
  echo "# pass $i: $pass" >> $out
  python $root/tools/pmc_summary.py $db ${PMC_FILTER-render} >> $out 2>&1
done
cat $out
