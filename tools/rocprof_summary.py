#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocprofv3 --kernel-trace --stats) into the text summary kept under profiles/.
Round 6: kernels that were launched with more than one grid z (the frames of a batched launch chain, gm_forward_deformed_batch_async)
are listed again per grid z - the name-level average mixes single-frame and K-frame launches."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
    lines = ["%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for n, calls, tot, avg, pct in rows:
        n = n if len(n) <= 90 else n[:87] + "..."
        lines.append("%-90s %8d %14d %12.0f %6.2f%%" % (n, calls, tot, avg, pct))
    try:
        per = list(c.execute("select name, grid_z, count(*), avg(duration) from kernels group by name, grid_z order by name, grid_z"))
        multi = {}
        for n, gz, cnt, avg in per:
            multi.setdefault(n, []).append((gz, cnt, avg))
        split = [(n, v) for n, v in multi.items() if len(v) > 1]
        if split:
            lines.append("")
            lines.append("launches by grid z (= frames per launch of the batched chain); avg in the unit of the durations above / 1000 if those are ns")
            lines.append("%-90s %6s %8s %12s" % ("kernel", "grid_z", "calls", "avg_us"))
            for n, v in sorted(split, key=lambda t: -sum(c_ * a for _, c_, a in t[1])):
                nn = n if len(n) <= 90 else n[:87] + "..."
                for gz, cnt, avg in v:
                    lines.append("%-90s %6d %8d %12.1f" % (nn, gz, cnt, avg / 1000.0))
    except sqlite3.Error as e:
        lines.append("(no per-grid split: %s)" % e)
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
