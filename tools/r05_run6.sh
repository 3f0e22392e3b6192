cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_topology.py tests/test_gpu_bg_route.py -q -m gpu -x > gpurun_out/r05_suite_3.txt 2>&1
tail -5 gpurun_out/r05_suite_3.txt
python tools/c5_densify_sweep.py "0.02,4" "0.02,12,1" "0.02,24,2" "0.02,48,3" > gpurun_out/r05_c5_densify_sweep.txt 2>gpurun_out/r05_c5_densify_sweep.err
cat gpurun_out/r05_c5_densify_sweep.txt; tail -3 gpurun_out/r05_c5_densify_sweep.err
