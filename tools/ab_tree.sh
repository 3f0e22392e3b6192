#!/bin/bash
# A/B of source TREES inside ONE gpurun call: tools/ab_tree.sh [fps|train] <dirA> <dirB> ... - each directory under tools/variants/ holds
# the csrc files of that variant (overlaid on gaussianmesh_amd/csrc, library rebuilt); three interleaved rounds.
cd ${GRAFT_REPO_ROOT:-.}
what=$1; shift
rm -rf /tmp/orig_csrc; cp -r gaussianmesh_amd/csrc /tmp/orig_csrc
for rep in 1 2 3; do
  for v in "$@"; do
    cp tools/variants/$v/* gaussianmesh_amd/csrc/
    (cd gaussianmesh_amd/csrc && make -B -j12 >/dev/null 2>&1) || echo "build failed: $v"
    if [ $what = fps ]; then
      python bench.py --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-fwd-bwd --no-c5 > gpurun_out/abt_$v.json 2> gpurun_out/abt_$v.err || tail -3 gpurun_out/abt_$v.err
      python - <<PY
import json
d=json.load(open("gpurun_out/abt_$v.json"))
print("$v", "%.0f" % d["value"], " ".join("%.0f" % x for x in d["repeats"]["frames_per_s"]), "1-stream %.4f" % d["single_stream"]["ms_per_frame"], d["stage_ms"], d["scene"])
PY
    else
      tools/fwd_bwd_once.sh $v | tail -1
      python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("   ", d["fwd_bwd"]["stage_ms"], d["fwd_bwd"]["scene"])
PY
    fi
  done
done
cp /tmp/orig_csrc/* gaussianmesh_amd/csrc/
(cd gaussianmesh_amd/csrc && make -B -j12 >/dev/null 2>&1)
