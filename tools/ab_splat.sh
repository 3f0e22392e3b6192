#!/bin/bash
# the 36-byte splat record (GM_SPLAT_STRIDE 9) against the 48-byte one (12): whole library rebuilt per variant, frame loop + forward/backward, interleaved
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for st in 9 12; do
    (cd gaussianmesh_amd/csrc && make clean >/dev/null && make HIPCC="/opt/rocm/bin/hipcc -DGM_SPLAT_STRIDE=$st" -j8 >/dev/null 2>&1) || echo "build failed: $st"
    echo -n "[stride $st] "; tools/ab_line.sh | cut -c1-180; echo -n "[stride $st] "; tools/fwd_bwd_line.sh | cut -c1-120
  done
done
(cd gaussianmesh_amd/csrc && make clean >/dev/null && make -j8 >/dev/null 2>&1)
