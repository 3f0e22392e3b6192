#!/bin/bash
# forward blend A/B on the GPU box: GM_FWD_V1=1 (round-2 kernel) vs the default, same library, same call
cd ${GRAFT_REPO_ROOT:-.}
for v in new v1 new v1; do
  if [ $v = v1 ]; then export GM_FWD_V1=1; else unset GM_FWD_V1; fi
  python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-fwd-bwd > gpurun_out/ab_fwd_$v.json 2> gpurun_out/ab_fwd_$v.err || tail -3 gpurun_out/ab_fwd_$v.err
  echo -n "$v "; python tools/show.py gpurun_out/ab_fwd_$v.json
done
