#!/usr/bin/env python3
"""rocpd database of a kernel trace -> duration of every paged_decode launch grouped by the kernel that ran right before it."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*$", "", n)[:70]
groups = {}
for i, (name, s, e) in enumerate(rows):
    if "paged_decode_kernel" not in name or i == 0:
        continue
    prev = rows[i - 1]
    gap = (s - prev[2]) / 1e3
    g = groups.setdefault(short(prev[0]), [])
    g.append(((e - s) / 1e3, gap))
for k, v in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    d = sorted(x[0] for x in v)
    gaps = sorted(x[1] for x in v)
    print(f"after {k:70s} n={len(v):4d}  attention median {d[len(d) // 2]:7.1f} us  min {d[0]:7.1f}  gap median {gaps[len(gaps) // 2]:6.1f} us")
