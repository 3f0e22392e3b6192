#!/usr/bin/env python
"""What fraction of the rows does densify_and_prune(0.0002) select at iteration 600 of the C5 loop, for a few settings of the student's
"under-reconstructed" rows (bench.C5_STUDENT_BIG: fraction, scale factor, opacity logit)?  The review asked for 1-2 %.
    python tools/c5_densify_sweep.py "0.02,4" "0.02,12,1" ..."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
for spec in sys.argv[1:]:
    os.environ["GM_C5_STUDENT"] = spec
    out = bench.c5_leg(600, 0, as_reference=True)
    q = out["viewspace_grad_at_first_densify"]
    print(spec, "ms/iter %.3f" % out["ms_per_iter"], "before densify %.3f" % (out["ms_per_iter_before_first_densify"] or 0), "densify iteration ms", out["densify_iterations_ms"],
          "rows", out["rows_after_densify"], "q50 %.2e q90 %.2e q99 %.2e q999 %.2e max %.2e over %.4f" % tuple(q[k] for k in ("q50", "q90", "q99", "q999", "max", "fraction_over_threshold")),
          "loss %.4f -> %.4f" % (out["loss_first"], out["loss_last"]), flush=True)
