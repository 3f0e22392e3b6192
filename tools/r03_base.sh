#!/bin/bash
# round-3 baseline on this round's box: driver command, default command, 4-stream kernel stats + SQ counters of the 4-stream loop
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03_base}
cd $root
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_20.json 2> gpurun_out/${tag}_20.err
python bench.py > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
tools/prof.sh ${tag}_4s --no-cpu-baseline --no-fwd-bwd --steps 100 --warmup 10 --repeats 0 > /dev/null
cd /tmp && export TMPDIR=/tmp
i=0
: > $root/gpurun_out/${tag}_4s_pmc.txt
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc4_$i
  timeout 240 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc4_$i -o p -- python $root/bench.py --no-cpu-baseline --no-fwd-bwd --steps 40 --warmup 10 --repeats 0 > $root/gpurun_out/${tag}_4s_pmc_bench_$i.json 2> /tmp/pmc4_$i.err
  db=$(find /tmp/pmc4_$i -name "*.db" | head -1)
  echo "# pass $i (4-stream bench loop): $pass" >> $root/gpurun_out/${tag}_4s_pmc.txt
  python $root/tools/pmc_summary.py $db "" >> $root/gpurun_out/${tag}_4s_pmc.txt 2>&1
  python $root/tools/rocprof_summary.py $db $root/gpurun_out/${tag}_4s_pmc_durations_$i.txt > /dev/null 2>&1
done
