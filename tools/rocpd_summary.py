#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` prints, and per-kernel PMC sums. Usage:
    rocpd_summary.py <results.db> [--pmc] > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    stats = {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        st = stats.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    tot = sum(v[1] for v in stats.values()) or 1.0
    print(f"# {sys.argv[1]}\n# kernel-trace summary (durations in microseconds)")
    print(f"{'kernel':110s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {v[0]:7d} {v[1]:12.1f} {v[1] / v[0]:10.2f} {v[2]:10.2f} {v[3]:10.2f} {100 * v[1] / tot:6.2f}")
    if "--pmc" in sys.argv:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        print("\n# PMC per kernel (sum over dispatches / dispatches)  columns:", ccols)
        q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"
        try:
            for kn, cn, sv, n in cur.execute(q):
                print(f"{short(kn):110s} {cn:14s} sum={sv:.6g} dispatches={n} per_dispatch={sv / n:.6g}")
        except sqlite3.OperationalError as ex:
            print("query failed:", ex)


if __name__ == "__main__":
    main()
