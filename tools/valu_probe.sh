cd ${GRAFT_REPO_ROOT:-.}
cp gaussianmesh_amd/csrc/gm_deform.hip /tmp/orig_deform.hip
for v in full noemit noslot; do
  cp tools/variants/$v/gm_deform.hip gaussianmesh_amd/csrc/; (cd gaussianmesh_amd/csrc && make >/dev/null 2>&1)
  GM_DEBUG_STOP_AFTER=deform PMC_TIMEOUT=120 PMC_BENCH_ARGS="--no-c5 --no-fwd-bwd" PMC_FILTER="deform_shade_kernel<true, true>" timeout 200 tools/pmc.sh probe_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" > /dev/null 2>&1
  echo "== $v"; grep -A4 "deform_shade" gpurun_out/probe_${v}_pmc.txt | head -5
done
cp /tmp/orig_deform.hip gaussianmesh_amd/csrc/gm_deform.hip
