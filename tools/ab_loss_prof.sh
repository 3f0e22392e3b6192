#!/bin/bash
# per-kernel durations (rocprofv3) of the loss kernels for source variants of gm_loss.hip under tools/variants/: tools/ab_loss_prof.sh v1 v2 ...
cd ${GRAFT_REPO_ROOT:-.}
root=$(pwd)
cp gaussianmesh_amd/csrc/gm_loss.hip /tmp/orig_loss.hip
for v in "$@"; do
  cp tools/variants/$v.hip gaussianmesh_amd/csrc/gm_loss.hip; (cd gaussianmesh_amd/csrc && make >/dev/null 2>&1)
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pl_$v && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pl_$v -o p -- python $root/tools/ssim_time.py > /dev/null 2>&1)
  echo "== $v"; db=$(find /tmp/pl_$v -name "*.db" | head -1)
  if [ -n "$db" ]; then timeout 60 python $root/tools/rocprof_summary.py $db /tmp/pl_$v.txt > /dev/null; grep -i ssim /tmp/pl_$v.txt | cut -c1-170; fi
done
cp /tmp/orig_loss.hip gaussianmesh_amd/csrc/gm_loss.hip; (cd gaussianmesh_amd/csrc && make >/dev/null 2>&1)
