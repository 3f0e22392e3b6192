#!/usr/bin/env python
"""Per-dispatch timeline of the LAST frame in a rocprofv3 kernel-trace .db: name, duration (us), gap to previous (us)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
# last frame = dispatches after the second-to-last deform/preprocess kernel
marks = [i for i, r in enumerate(rows) if "deform" in r[0] or "preprocess_fwd" in r[0]]
key = sys.argv[2] if len(sys.argv) > 2 else "deform"
marks = [i for i, r in enumerate(rows) if key in r[0]]
start = marks[-1] if marks else 0
prev_end = None
tot = 0
for n, s, e in rows[start:]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-70s %9.1f us   gap %7.1f" % (n[:70], (e - s) / 1e3, gap))
    tot += (e - s) / 1e3
    prev_end = e
print("sum of kernel time %.1f us, span %.1f us" % (tot, (rows[-1][2] - rows[start][1]) / 1e3))
