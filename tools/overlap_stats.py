#!/usr/bin/env python
"""Concurrency statistics of a rocprofv3 kernel-trace .db over a window of steady-state dispatches: GPU busy fraction
(union of kernel intervals), mean number of kernels in flight, and per-kernel mean duration inside the window."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
key = sys.argv[2] if len(sys.argv) > 2 else "deform_shade"
marks = [i for i, r in enumerate(rows) if key in r[0]]
lo, hi = marks[len(marks) // 4], marks[3 * len(marks) // 4]          # middle half of the frames
win = rows[lo:hi]
t0, t1 = win[0][1], max(r[2] for r in win)
ev = sorted([(s, 1) for _, s, e in win] + [(e, -1) for _, s, e in win])
busy = 0; area = 0; depth = 0; prev = t0
for t, d in ev:
    if depth > 0:
        busy += t - prev
    area += depth * (t - prev)
    depth += d; prev = t
frames = (hi - lo) and len([1 for r in win if key in r[0]])
print("window %.3f ms, %d frames -> %.3f ms/frame; GPU busy %.1f%%; mean kernels in flight %.2f" % (
    (t1 - t0) / 1e6, frames, (t1 - t0) / 1e6 / max(frames, 1), 100.0 * busy / (t1 - t0), area / max(busy, 1)))
acc = collections.defaultdict(lambda: [0, 0])
for n, s, e in win:
    acc[n[:60]][0] += e - s; acc[n[:60]][1] += 1
for n, (tot, cnt) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-60s %6d calls  mean %8.1f us   per frame %8.1f us" % (n, cnt, tot / cnt / 1e3, tot / 1e3 / max(frames, 1)))
