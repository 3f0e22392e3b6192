# usage (GPU box, repo root): tools/prof_all.sh <tag> ; copy the gpurun_out/<tag>_* files you keep into profiles/
tag=${1:-r05_v1}
set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_20steps.json 2>/dev/null
python bench.py --streams 1 --exact-count --no-cpu-baseline --no-c5 > gpurun_out/${tag}_bench_1stream.json 2>/dev/null
python bench.py --backward-state --no-cpu-baseline --no-fwd-bwd > gpurun_out/${tag}_bench_backward_state.json 2>/dev/null
tools/prof.sh ${tag} --no-cpu-baseline --no-fwd-bwd > /dev/null
tools/prof.sh ${tag}_1stream --no-cpu-baseline --no-c5 --streams 1 --exact-count > /dev/null
PMC_BENCH_ARGS="--no-c5" PMC_FILTER="" tools/pmc.sh ${tag}_hbm_traffic FETCH_SIZE WRITE_SIZE > /dev/null
PMC_BENCH_ARGS="--no-c5" PMC_FILTER="" tools/pmc.sh ${tag}_inst_mix "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" > /dev/null
python tools/hbm_traffic.py gpurun_out/${tag}_hbm_traffic_pmc.txt gpurun_out/${tag}_hbm_traffic.json 1000000 1920 1080 > /dev/null
(cd tools && python inst_mix.py ../gpurun_out/${tag}_inst_mix_pmc.txt ../gpurun_out/${tag}_inst_mix.json 1000000 1920 1080 > /dev/null)
python bench.py --config c5 > gpurun_out/${tag}_bench_c5.json 2>/dev/null
tools/prof.sh ${tag}_c5 --config c5 --steps 100 > /dev/null
python tools/pipeline_trace.py > gpurun_out/${tag}_pipeline_trace.txt 2>/dev/null
ls -la gpurun_out | tail -15
