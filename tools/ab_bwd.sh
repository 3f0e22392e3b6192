#!/bin/bash
# tools/ab_bwd.sh: forward+backward of bench.py (C3 cloud, C2) with the product backward blend (exponents from the matrix core) and with the
# verification build (round 3's per-pixel exponent), in one call on one box.
for v in 0 1 0 1; do
python - <<PY
import ctypes, json, sys, io, contextlib
from gaussianmesh_amd import _lib
ctypes.CDLL(_lib.lib()._name).gm_debug_backward_exact_exponent($v)
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-c5", "--steps", "40", "--warmup", "10", "--repeats", "0"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
f = d["fwd_bwd"]
print("exact=$v", "fwd_bwd %.4f" % f["ms_per_iter"], "render_bwd %.4f" % f["stage_ms"]["render_bwd"], "c2 %.4f" % f["c2_500k_ms_per_iter"], "train %.4f" % f["ms_per_training_iteration"],
      "render %.4f" % f["stage_ms"]["render"], "fps %.0f" % d["value"], flush=True)
PY
done
