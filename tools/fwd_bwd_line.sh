#!/bin/bash
# one line: forward + backward, training iteration and C2 of `python bench.py --no-cpu-baseline --no-c5 --steps 50` (for tools/ab_flags.sh)
python bench.py --no-cpu-baseline --no-c5 --steps 50 --warmup 10 --repeats 0 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.readline()); fb=d["fwd_bwd"]
print("fwd_bwd %.4f train_iter %.4f c2 %.4f stage %s" % (fb["ms_per_iter"], fb["ms_per_training_iteration"], fb["c2_500k_ms_per_iter"], json.dumps(fb["stage_ms"])))'
