cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB_BENCH_ARGS="--steps 200 --warmup 10 --repeats 2 --no-variants"
tools/ab_flags.sh gm_render tools/ab_line.sh "-DGM_FWD_MASKS=0" "-DGM_FWD_MASKS=1" > gpurun_out/r05_ab_fwd_masks.txt 2>&1
cat gpurun_out/r05_ab_fwd_masks.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/r05_suite_2.txt 2>&1
tail -3 gpurun_out/r05_suite_2.txt
