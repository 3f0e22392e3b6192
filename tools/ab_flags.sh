#!/bin/bash
# A/B of compile-time variants of csrc/<file> inside ONE gpurun call: tools/ab_flags.sh <stem> <script> "<flags A>" "<flags B>" ...
# rebuilds the library with EXTRA_<stem>="<base flags> <flags X>" and runs <script> <tag>; every variant twice, interleaved
cd ${GRAFT_REPO_ROOT:-.}
stem=$1; script=$2; shift; shift
base=$(grep "^EXTRA_$stem" gaussianmesh_amd/csrc/Makefile | sed "s/^EXTRA_$stem = //")
for rep in 1 2; do
  i=0
  for fl in "$@"; do
    i=$((i+1))
    (cd gaussianmesh_amd/csrc && touch $stem.hip && make EXTRA_$stem="$base $fl" >/dev/null 2>&1) || echo "build failed: $fl"
    echo "[$fl]"; $script v$i | tail -${AB_LINES:-1}
  done
done
(cd gaussianmesh_amd/csrc && touch $stem.hip && make >/dev/null 2>&1)
