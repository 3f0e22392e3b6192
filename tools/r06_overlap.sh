#!/bin/bash
# GPU busy fraction / kernels in flight in the middle of the batched loop (K = 4, four streams, 300 steps)
cd $GRAFT_REPO_ROOT
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ovl
rocprofv3 --kernel-trace -d /tmp/ovl -o p -- python $root/bench.py --steps 300 --warmup 20 --repeats 0 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants > /dev/null 2>&1
db=$(find /tmp/ovl -name "*.db" | head -1)
python $root/tools/overlap_stats.py $db deform_shade_pre_batch | tee $root/gpurun_out/r06_overlap_k4.txt
