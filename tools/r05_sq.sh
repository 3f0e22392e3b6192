#!/bin/bash
# SQ counter passes over the two blend kernels (round 5, review item 2): what bounds them - vector issue, waits, LDS?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r05_sq_counter_names.txt)
wc -l gpurun_out/r05_sq_counter_names.txt
have() { grep -qx "$1" gpurun_out/r05_sq_counter_names.txt && echo -n "$1 "; }
P1="$(have SQ_INSTS_VALU)$(have SQ_ACTIVE_INST_VALU)$(have SQ_BUSY_CYCLES)$(have SQ_WAVE_CYCLES)$(have SQ_WAIT_INST_ANY)$(have SQ_WAIT_ANY)$(have SQ_ACTIVE_INST_ANY)$(have SQ_INSTS_SALU)"
P2="$(have SQ_INSTS_LDS)$(have SQ_ACTIVE_INST_LDS)$(have SQ_LDS_BANK_CONFLICT)$(have SQ_INSTS_MFMA)$(have SQ_VALU_MFMA_BUSY_CYCLES)$(have SQ_WAIT_INST_LDS)$(have SQ_ACTIVE_INST_SCA)$(have SQ_INSTS_VALU_TRANS)"
P3="$(have SQ_INST_CYCLES_VALU)$(have SQ_INSTS_VALU_TRANS)$(have SQ_THREAD_CYCLES_VALU)$(have SQ_WAVES)$(have SQ_INST_CYCLES_SALU)$(have SQ_LEVEL_WAVES)$(have SQ_INSTS_VMEM)$(have SQ_ACTIVE_INST_VMEM)"
echo "P1=$P1"; echo "P2=$P2"; echo "P3=$P3"
PMC_TIMEOUT=400 PMC_BENCH_ARGS="--no-c5" PMC_FILTER="render" tools/pmc.sh r05_blend_sq "$P1" "$P2" "$P3" > /dev/null 2>&1
cat gpurun_out/r05_blend_sq_pmc.txt
