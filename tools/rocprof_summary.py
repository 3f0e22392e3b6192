#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocprofv3 --kernel-trace --stats) into the text summary kept under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
    lines = ["%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for n, calls, tot, avg, pct in rows:
        n = n if len(n) <= 90 else n[:87] + "..."
        lines.append("%-90s %8d %14d %12.0f %6.2f%%" % (n, calls, tot, avg, pct))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
