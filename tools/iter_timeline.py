#!/usr/bin/env python
"""Kernel timeline of the LAST complete training iteration in a rocprofv3 kernel-trace .db (between two mesh_activate_fwd)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "mesh_activate_fwd" in r[0]]
a, b = marks[-2], marks[-1]
prev = None; tot = 0
for n, s, e in rows[a:b]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-84s %8.1f us  gap %7.1f" % (n[:84], (e - s) / 1e3, gap))
    tot += (e - s) / 1e3; prev = e
print("kernels %d, sum %.1f us, span %.1f us" % (b - a, tot, (rows[b][1] - rows[a][1]) / 1e3))
