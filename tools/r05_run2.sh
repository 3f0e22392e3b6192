cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/valu_probe && ./probe 2048 22 > ../../gpurun_out/r05_valu_probe_v2.txt 2>&1)
tail -50 gpurun_out/r05_valu_probe_v2.txt
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r05_suite_1.txt 2>&1
tail -5 gpurun_out/r05_suite_1.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-c5 > gpurun_out/r05_v1_bench_20steps.json 2> gpurun_out/r05_v1_bench_20steps.err
python tools/show.py gpurun_out/r05_v1_bench_20steps.json
