#!/bin/bash
# A/B of whole-library compile flags in ONE gpurun call: tools/ab_all.sh "<flags A>" "<flags B>" ...: rebuild everything with
# CXXFLAGS += flags, run a 100-step bench (+ 2 repeats) twice per variant, interleaved
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for fl in "$@"; do
    (cd gaussianmesh_amd/csrc && make clean >/dev/null && make HIPCC="/opt/rocm/bin/hipcc $fl" -j8 >/dev/null 2>&1) || echo "build failed: $fl"
    echo -n "[$fl] "; tools/fwd_once.sh v | tail -1
  done
done
(cd gaussianmesh_amd/csrc && make clean >/dev/null && make -j8 >/dev/null 2>&1)
