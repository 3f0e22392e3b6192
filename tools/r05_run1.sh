cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/valu_probe && ./probe 2048 > ../../gpurun_out/r05_valu_probe.txt 2>&1)
timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py -x -q -s -m gpu > gpurun_out/r05_fuzz_parity_test.txt 2>&1
tail -5 gpurun_out/r05_fuzz_parity_test.txt
timeout 600 python tools/needle_truth.py 120 126 138 162 168 216 > gpurun_out/r05_needle_truth.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_v0_bench_20steps.json 2> gpurun_out/r05_v0_bench_20steps.err
tail -c 600 gpurun_out/r05_v0_bench_20steps.json
