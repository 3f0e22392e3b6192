#!/bin/bash
# round 6: SQ / memory counter passes over the kernels of the BATCHED loop (K = 4 on one stream): what bounds the fused pass now?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r06_batch_sq}
flt=${2:-deform_shade_pre_batch}
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r06_counter_names.txt)
have() { grep -qx "$1" gpurun_out/r06_counter_names.txt && echo -n "$1 "; }
P1="$(have SQ_INSTS_VALU)$(have SQ_ACTIVE_INST_VALU)$(have SQ_BUSY_CYCLES)$(have SQ_WAVE_CYCLES)$(have SQ_WAIT_INST_ANY)$(have SQ_WAIT_ANY)$(have SQ_ACTIVE_INST_ANY)$(have SQ_INSTS_SALU)"
P2="$(have SQ_INSTS_LDS)$(have SQ_ACTIVE_INST_LDS)$(have SQ_LDS_BANK_CONFLICT)$(have SQ_WAIT_INST_LDS)$(have SQ_INSTS_VMEM)$(have SQ_ACTIVE_INST_VMEM)$(have SQ_INSTS_VALU_TRANS)$(have SQ_WAVES)"
P3="$(have TCP_TOTAL_CACHE_ACCESSES_sum)$(have TCP_TCC_READ_REQ_sum)$(have TCC_HIT_sum)$(have TCC_MISS_sum)$(have TCC_REQ_sum)$(have TCP_TOTAL_ACCESSES_sum)"
echo "P1=$P1"; echo "P2=$P2"; echo "P3=$P3"
PMC_TIMEOUT=400 PMC_BENCH_ARGS="--no-fwd-bwd --no-c5 --no-variants --repeats 0" PMC_MODE_ARGS="--streams 1 --frames-per-launch 4" PMC_STEPS=8 PMC_FILTER="$flt" tools/pmc.sh $tag "$P1" "$P2" "$P3" > /dev/null 2>&1
cat gpurun_out/${tag}_pmc.txt
