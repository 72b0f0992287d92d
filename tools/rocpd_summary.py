#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: tools/rocpd_summary.py results.db [> profiles/rNN_name_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-64s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for n, c, t, a, mn, mx in rows:
    n = n.replace("void ", "").replace("ptx::", "").replace("(anonymous namespace)::", "")
    n = (n.split("<")[0] + ("<" + n.split("<", 1)[1].split(">(")[0] + ">" if "<" in n else "")).split("(")[0][:64]
    print("%-64s %8d %14d %12.0f %12d %12d %6.2f%%" % (n, c, t, a, mn, mx, 100.0 * t / tot))

# optional: --timeline N  prints the last N dispatches (start offset, duration, gap to the previous end) in us
if "--timeline" in sys.argv:
    n_last = int(sys.argv[sys.argv.index("--timeline") + 1])
    tl = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()[-n_last:]
    t0, prev = tl[0][1], None
    print("\n%-56s %10s %9s %9s" % ("dispatch", "start_us", "dur_us", "gap_us"))
    for n, st, en in tl:
        print("%-56s %10.1f %9.1f %9.1f" % (n.replace("void ", "").replace("ptx::", "").split("(")[0][:56], (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3 if prev else 0.0))
        prev = en
