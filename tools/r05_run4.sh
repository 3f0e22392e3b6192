cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB_LINES=2 tools/ab_flags.sh gm_render tools/fwd_bwd_once.sh "-DGM_BWD_PREFETCH=1" "-DGM_BWD_PREFETCH=2" > gpurun_out/r05_ab_bwd_sets2.txt 2>&1
cat gpurun_out/r05_ab_bwd_sets2.txt
