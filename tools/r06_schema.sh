#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sch && rocprofv3 --kernel-trace --stats -d /tmp/sch -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --repeats 0 --gaussians 50000 --width 320 --height 200 --no-cpu-baseline --no-fwd-bwd --no-c5 --no-variants > /dev/null 2>&1
db=$(find /tmp/sch -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$db")
for (n,t) in c.execute("select name, type from sqlite_master where type in ('table','view')"):
    if any(k in n.lower() for k in ("kernel","dispatch","counter")):
        cols=[d[0] for d in c.execute("select * from %s limit 1" % n).description]
        print(t, n, cols)
PY
