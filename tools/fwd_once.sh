#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-fwd-bwd > gpurun_out/fwd_$1.json 2> gpurun_out/fwd_$1.err || tail -3 gpurun_out/fwd_$1.err
python tools/show.py gpurun_out/fwd_$1.json
