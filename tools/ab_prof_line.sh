#!/bin/bash
# one line per run for tools/ab_flags.sh: rocprof average of the kernels matching $AB_KERNEL in a single-stream bench run
tag=$1
PROF_LINES=40 tools/prof.sh ab_$tag --no-cpu-baseline --no-c5 --no-fwd-bwd --streams 1 --exact-count $AB_BENCH_ARGS | grep -E "${AB_KERNEL:-deform_shade}" | awk '{print $(NF-3), $(NF-1), $1, $2, $3, $4}' | head -${AB_KLINES:-2} | tr '\n' ';'
echo
