# usage (GPU box, repo root): tools/prof_all.sh ; copy the gpurun_out/r02_v14_* files you keep into profiles/
set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_v14_bench.json 2> gpurun_out/r02_v14_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_v14_bench_20steps.json 2>/dev/null
python bench.py --streams 1 --exact-count --no-cpu-baseline > gpurun_out/r02_v14_bench_1stream.json 2>/dev/null
python bench.py --backward-state --no-cpu-baseline --no-fwd-bwd > gpurun_out/r02_v14_bench_backward_state.json 2>/dev/null
tools/prof.sh r02_v14 --no-cpu-baseline --no-fwd-bwd > /dev/null
tools/prof.sh r02_v14_1stream --no-cpu-baseline --streams 1 --exact-count > /dev/null
PMC_FILTER="" tools/pmc.sh r02_v14_hbm_traffic FETCH_SIZE WRITE_SIZE > /dev/null
PMC_FILTER="" tools/pmc.sh r02_v14_inst_mix "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" > /dev/null
python bench.py --config c5 > gpurun_out/r02_v14_bench_c5.json 2>/dev/null
ls -la gpurun_out | tail -15
