#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd schema) into the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints: calls, total/avg/min/max duration, % of GPU time.
Usage: scripts/rocprof_summary.py results.db [> profiles/xxx.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-72s %8s %14s %12s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for n, cnt, s, a, mn, mx in rows:
        print("%-72s %8d %14d %12.0f %10d %10d %6.2f%%" % (n[:72], cnt, s, a, mn, mx, 100.0 * s / tot))


if __name__ == "__main__":
    main(sys.argv[1])
