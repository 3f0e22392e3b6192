# usage (GPU box): tools/ab_ahead.sh ; --begin-ahead 1..4 at 300 and at 20 steps, two interleaved rounds
pick='import json,sys
d=json.loads(sys.stdin.readline()); r=d["repeats"]["frames_per_s"]; import statistics; print(sys.argv[1], "value %.0f" % d["value"], "median of repeats %.0f" % statistics.median(r), "repeats", " ".join("%.0f"%x for x in r))'
for rep in 1 2; do
for a in 2 3 1 4; do
  python bench.py --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-fwd-bwd --no-c5 --begin-ahead $a 2>/dev/null | python -c "$pick" "[300 steps, ahead $a]"
  python bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-fwd-bwd --no-c5 --begin-ahead $a 2>/dev/null | python -c "$pick" "[ 20 steps, ahead $a]"
done
done
