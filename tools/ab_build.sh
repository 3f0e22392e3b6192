#!/bin/bash
# A/B of whole-library compile flags in ONE gpurun call: tools/ab_build.sh "<flags A>" "<flags B>" ... : rebuild with hipcc + flags, run
# `python bench.py $AB_ARGS` (default: the batched loop, 300 steps, 2 repeats, nothing else) twice per variant, interleaved.  Leaves the default build.
cd ${GRAFT_REPO_ROOT:-.}
args=${AB_ARGS:---steps 300 --warmup 20 --repeats 2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants}
for rep in 1 2; do
  for fl in "$@"; do
    (cd gaussianmesh_amd/csrc && make clean >/dev/null && make HIPCC="/opt/rocm/bin/hipcc $fl" -j8 >/dev/null 2>&1) || echo "build failed: $fl"
    echo -n "[$fl] "
    python bench.py $args 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("fps %.1f repeats %s R %s stage_ms %s" % (d["value"], d.get("repeats",{}).get("frames_per_s"), d.get("scene",{}).get("R"), json.dumps(d.get("stage_ms"))))'
  done
done
(cd gaussianmesh_amd/csrc && make clean >/dev/null && make -j8 >/dev/null 2>&1)
