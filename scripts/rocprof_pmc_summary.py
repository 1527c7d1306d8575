#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 results .db (rocpd schema).
Usage: scripts/rocprof_pmc_summary.py results.db [...]  -> table of mean counter value per launch.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B read
requests as 64 B, so the HBM-read estimate printed here is 2 x FETCH_SIZE (MI355X_MICROARCH.md,
section HBM); WRITE_SIZE is uncalibrated and printed as is."""
import sqlite3
import sys


def main(paths):
    for path in paths:
        c = sqlite3.connect(path)
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        print("# %s" % path)
        print("%-64s %-26s %8s %16s" % ("kernel", "counter", "launches", "mean/launch"))
        for k, n, cnt, avg in rows:
            if "mfm::" not in k:
                continue
            extra = ""
            if n == "FETCH_SIZE":
                extra = "   -> HBM read ~ %.1f KB (x2 gfx950 correction)" % (2.0 * avg)
            if n == "WRITE_SIZE":
                extra = "   -> HBM write ~ %.1f KB" % avg
            print("%-64s %-26s %8d %16.1f%s" % (k[:64], n, cnt, avg, extra))


if __name__ == "__main__":
    main(sys.argv[1:])
