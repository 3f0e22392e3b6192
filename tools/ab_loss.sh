cd ${GRAFT_REPO_ROOT:-.}
cp gaussianmesh_amd/csrc/gm_loss.hip /tmp/orig_loss.hip
for rep in 1 2; do for v in ${VARIANTS:-loss_base loss_alias}; do
  cp tools/variants/$v.hip gaussianmesh_amd/csrc/gm_loss.hip; (cd gaussianmesh_amd/csrc && make >/dev/null 2>&1)
  echo -n "$v: "; python tools/ssim_time.py 2>/dev/null | tr '\n' ' '; echo
done; done
cp /tmp/orig_loss.hip gaussianmesh_amd/csrc/gm_loss.hip; (cd gaussianmesh_amd/csrc && make >/dev/null 2>&1)
