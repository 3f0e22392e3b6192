cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --config c5 > gpurun_out/r05_c5_r05scene.json 2>/dev/null
python bench.py --config c5 --c5-scene r04 > gpurun_out/r05_c5_r04scene.json 2>/dev/null
python - <<'PY'
import json
for f in ("r05_c5_r05scene", "r05_c5_r04scene"):
    d = json.load(open("gpurun_out/%s.json" % f))["c5"]
    print(f, "ms/iter %.3f before %.3f densify ms %s rows %s loss %.4f->%.4f redone %d frac_over %.5f" % (d["ms_per_iter"], d["ms_per_iter_before_first_densify"], d["densify_iterations_ms"], d["rows_after_densify"], d["loss_first"], d["loss_last"], d["iterations_redone"], d["viewspace_grad_at_first_densify"]["fraction_over_threshold"]), d["forced_densify"])
PY
tools/prof.sh r05_c5 --config c5 --steps 100 > /dev/null 2>&1
head -40 gpurun_out/r05_c5_kernel_stats.txt
