cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB_LINES=2 tools/ab_flags.sh gm_render tools/fwd_bwd_once.sh "-DGM_BWD_MATRIX=0" "-DGM_BWD_MATRIX=1" "-DGM_BWD_MATRIX=0 -DGM_BWD_SETS=2" > gpurun_out/r05_ab_bwd_matrix.txt 2>&1
cat gpurun_out/r05_ab_bwd_matrix.txt
