#!/bin/bash
# round 6: instruction counts / busy cycles of the scatter kernels under both ranking schemes (GM_SCATTER_RANK 0 | 1), single-frame launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES"
for v in 0 1; do
  (cd gaussianmesh_amd/csrc && make clean >/dev/null && make HIPCC="/opt/rocm/bin/hipcc -DGM_SCATTER_RANK=$v" -j8 >/dev/null 2>&1) || echo "build failed"
  PMC_TIMEOUT=300 PMC_BENCH_ARGS="--no-fwd-bwd --no-c5 --no-variants --repeats 0" PMC_MODE_ARGS="--streams 1 --frames-per-launch 1" PMC_STEPS=8 PMC_FILTER="bk_scatter" tools/pmc.sh r06_scatter_rank$v "$P1" "$P2" > /dev/null 2>&1
  echo "== GM_SCATTER_RANK=$v"; cat gpurun_out/r06_scatter_rank${v}_pmc.txt
done
(cd gaussianmesh_amd/csrc && make clean >/dev/null && make -j8 >/dev/null 2>&1)
