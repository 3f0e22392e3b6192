#!/bin/bash
# round 6: kernel table of the batched loop on ONE stream (per-launch durations without overlap) and on two, + the refusal test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_c}
timeout 600 python -m pytest tests/test_gpu_batch.py -q -x > gpurun_out/${tag}_batch_tests.txt 2>&1; tail -3 gpurun_out/${tag}_batch_tests.txt
PROF_LINES=16 bash tools/prof.sh ${tag}_k4_s1 --steps 200 --warmup 20 --repeats 0 --frames-per-launch 4 --streams 1 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants
PROF_LINES=16 bash tools/prof.sh ${tag}_k4_s2 --steps 200 --warmup 20 --repeats 0 --frames-per-launch 4 --streams 2 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants
PROF_LINES=16 bash tools/prof.sh ${tag}_k1_s1 --steps 200 --warmup 20 --repeats 0 --frames-per-launch 1 --streams 1 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants
