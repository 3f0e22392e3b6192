#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> <bench args...>
# rocprofv3 --kernel-trace --stats of `python bench.py <args>`; summary -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $root/bench.py "$@" > $root/gpurun_out/${tag}_profiled_bench.json 2> /tmp/prof_$tag.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $root/tools/rocprof_summary.py $db $root/gpurun_out/${tag}_kernel_stats.txt > /dev/null
head -${PROF_LINES:-25} $root/gpurun_out/${tag}_kernel_stats.txt
