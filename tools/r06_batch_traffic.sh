#!/bin/bash
# round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of the batched loop, K = 4 on one stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PMC_TIMEOUT=400 PMC_BENCH_ARGS="--no-fwd-bwd --no-c5 --no-variants --repeats 0" PMC_MODE_ARGS="--streams 1 --frames-per-launch 4" PMC_STEPS=8 PMC_FILTER="" tools/pmc.sh r06_batch_hbm_traffic FETCH_SIZE WRITE_SIZE > /dev/null 2>&1
grep -A2 -E "pre_batch|render_fwd_kernel<false|bucket_sort|duplicate_kernel<1, 2>|bk_scatter|bk_hist|mesh_rs" gpurun_out/r06_batch_hbm_traffic_pmc.txt | head -80
