#!/bin/bash
# A/B of kernel source variants on the pipelined forward loop inside ONE gpurun call: tools/ab_fps.sh <file.hip> <variantA> <variantB> ...
# (variants under tools/variants/); three interleaved rounds, each a 300-step region + 3 repeats; prints frames/s of every region
cd ${GRAFT_REPO_ROOT:-.}
f=$1; shift
cp gaussianmesh_amd/csrc/$f /tmp/orig_$f
for rep in 1 2 3; do
  for v in "$@"; do
    cp tools/variants/$v gaussianmesh_amd/csrc/$f
    (cd gaussianmesh_amd/csrc && make -B -j12 >/dev/null 2>&1)
    python bench.py --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-fwd-bwd --no-c5 > gpurun_out/abf_$v.json 2> gpurun_out/abf_$v.err || tail -3 gpurun_out/abf_$v.err
    python - <<PY
import json
d=json.load(open("gpurun_out/abf_$v.json"))
print("$v", "%.0f" % d["value"], " ".join("%.0f" % x for x in d["repeats"]["frames_per_s"]), "1-stream %.4f" % d["single_stream"]["ms_per_frame"])
PY
  done
done
cp /tmp/orig_$f gaussianmesh_amd/csrc/$f
(cd gaussianmesh_amd/csrc && make -B -j12 >/dev/null 2>&1)
