#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc run (.db): per kernel name, mean of each counter over dispatches."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]); sys.exit(0)
cols = [d[0] for d in c.execute(f"select * from {view} limit 1").description]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in c.execute(f"select * from {view}"):
    r = dict(zip(cols, row))
    key = r.get("kernel_name") or r.get("name")
    gz = r.get("grid_size_z")
    if gz is not None and int(gz) > 1:                 # launches over K frames (grid z) are kept apart from single-frame launches of the same kernel
        key = "%s   [grid z = %d]" % (key[:100], int(gz))
    acc[key][r["counter_name"]].append(r["value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, d in acc.items():
    if flt and flt not in k: continue
    print(k[:130])
    for n, v in sorted(d.items()):
        print("   %-32s mean %.4g  (n=%d)" % (n, sum(v) / len(v), len(v)))
