#!/bin/bash
# usage (GPU box): tools/sweep_env.sh VAR "v1 v2 ..." [bench args]: one short bench run per value of an environment variable
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v timeout 300 python bench.py --no-cpu-baseline --no-fwd-bwd --repeats 1 "$@" > /tmp/sweep.json 2>/dev/null
  python - "$var=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print(sys.argv[1], round(d["value"], 1), d["repeats"]["frames_per_s"], round(d["single_stream"]["ms_per_frame"], 4), d["stage_ms"])
PY
done
