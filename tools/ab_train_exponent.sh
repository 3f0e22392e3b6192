pick='import json,sys
d=json.loads(sys.stdin.readline()); fb=d["fwd_bwd"]
print(sys.argv[1], "fwd_bwd %.4f" % fb["ms_per_iter"], "train_iter %.4f" % fb["ms_per_training_iteration"], "c2 %.4f" % fb["c2_500k_ms_per_iter"], "c5 %.3f" % d["c5"]["ms_per_iter"], "c5_fixed %.3f" % d["c5_fixed"]["ms_per_iter"], "render_ms", fb["stage_ms"]["render"])'
for r in 1 2; do
python tools/with_debug.py none bench.py --no-cpu-baseline 2>/dev/null | python -c "$pick" matrix_forward
python tools/with_debug.py fwd_exact bench.py --no-cpu-baseline 2>/dev/null | python -c "$pick" exact_forward
done
