#!/bin/bash
# A/B of kernel source variants inside ONE gpurun call (boxes differ by several percent): tools/ab_variants.sh <file.hip> <variantA> <variantB> ...
# each variant is a source file under gpurun_out/variants/; it is copied over csrc/<file.hip>, the library rebuilt, tools/fwd_bwd_once.sh run.
cd ${GRAFT_REPO_ROOT:-.}
f=$1; shift
cp gaussianmesh_amd/csrc/$f /tmp/orig_$f
for v in "$@" "$@"; do
  cp tools/variants/$v gaussianmesh_amd/csrc/$f
  (cd gaussianmesh_amd/csrc && make >/dev/null 2>&1)
  tools/fwd_bwd_once.sh $v | tail -1
done
cp /tmp/orig_$f gaussianmesh_amd/csrc/$f
