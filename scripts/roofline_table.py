#!/usr/bin/env python3
"""Per-kernel roofline table from one profile round (scripts/profile_round6.sh):
  kernel-trace db  -> average duration per launch
  pmc_fetch/write  -> HBM bytes per launch (read = 2 x FETCH_SIZE KiB on gfx950, write = WRITE_SIZE KiB)
  pmc_sq           -> SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU / MFMA / LDS per launch
Prints HBM GB/s against the 8 TB/s peak and the MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the
SIMDs) / (duration x 2.4 GHz x 1024 SIMDs); the clock under load is lower (DVFS), so this slightly under-states it.
Usage: scripts/roofline_table.py <dir with ktrace_results.db pmc_*_results.db>"""
import glob
import os
import sqlite3
import sys

HBM_PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def avg_by_kernel(db, sql):
    c = sqlite3.connect(db)
    out = {}
    for name, val in c.execute(sql):
        out.setdefault(name, []).append(val)
    return {k: sum(v) / len(v) for k, v in out.items()}


def counters(db):
    c = sqlite3.connect(db)
    out = {}
    for name, cn, val in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(name, {})[cn] = val
    return out


def main(d):
    kt = glob.glob(os.path.join(d, "ktrace*.db"))[0]
    dur = avg_by_kernel(kt, "select name, end - start from kernels")
    fetch = counters(glob.glob(os.path.join(d, "pmc_fetch*.db"))[0])
    write = counters(glob.glob(os.path.join(d, "pmc_write*.db"))[0])
    sq = counters(glob.glob(os.path.join(d, "pmc_sq*.db"))[0])
    print("%-58s %8s %9s %9s %8s %7s %10s %10s %9s" % ("kernel", "us", "HBM rd MB", "HBM wr MB", "GB/s", "%peak", "VALU inst", "MFMA inst", "MFMA util"))
    for k in sorted(dur, key=lambda n: -dur[n]):
        if "mfm::" not in k:
            continue
        us = dur[k] / 1e3
        rd = 2.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 / 1e6
        wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
        gbs = (rd + wr) * 1e6 / (us * 1e-6) / 1e9 if us else 0.0
        s = sq.get(k, {})
        busy = us * 1e-6 * 2.4e9 * 1024
        mb = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        print("%-58s %8.2f %9.2f %9.2f %8.1f %6.2f%% %10.0f %10.0f %8.1f%%" % (
            k[:58], us, rd, wr, gbs, 100 * gbs / HBM_PEAK, s.get("SQ_INSTS_VALU", 0.0), s.get("SQ_INSTS_MFMA", 0.0),
            100 * mb / busy if busy else 0.0))


if __name__ == "__main__":
    main(sys.argv[1])
