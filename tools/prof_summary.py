#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) capture into the per-kernel table committed under profiles/.
    python tools/prof_summary.py gpurun_out/prof/trace_results.db profiles/r01_kernel_stats.csv
    python tools/prof_summary.py gpurun_out/prof_pmc/pmc_results.db profiles/r01_pmc_fetch.csv --pmc"""
import csv
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name"))
    tot = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    return [dict(kernel=r[0], calls=r[1], total_ms=r[2] / 1e6, avg_us=r[3] / 1e3, min_us=r[4] / 1e3, max_us=r[5] / 1e3,
                 pct=100.0 * r[2] / tot) for r in rows]


def pmc_stats(db):
    """per kernel: dispatches, average raw counter value, and for FETCH_SIZE the HBM bytes per launch
    with the gfx950 correction (the counter tallies 128-B requests at 64 B: x2 for wide coalesced
    streaming reads, guides/MI355X_MICROARCH.md §HBM)."""
    cur = sqlite3.connect(db).cursor()
    out = []
    q = ("select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration) "
         "from pmc_events group by name, counter_name")
    for k, c, n, avg, lo, hi, dur in cur.execute(q):
        row = dict(kernel=k, counter=c, dispatches=n, avg_value=avg, min_value=lo, max_value=hi, avg_us=dur / 1e3)
        if c == "FETCH_SIZE":
            row["hbm_read_bytes_per_launch_x2"] = avg * 1024 * 2
        out.append(row)
    out.sort(key=lambda r: -r["avg_value"] * r["dispatches"])
    return out


if __name__ == "__main__":
    db, dst = sys.argv[1], sys.argv[2]
    rows = pmc_stats(db) if "--pmc" in sys.argv else kernel_stats(db)
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            w.writerow({k: (f"{v:.3f}" if isinstance(v, float) else v) for k, v in r.items()})
    for r in rows[:16]:
        print(r)
