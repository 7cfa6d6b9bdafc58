#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a per-kernel stats table
(calls, total / average / min / max duration).  Usage: summarize_rocpd.py results.db > summary.md"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for n, c, t, a, mn, mx in rows:
    print(f"| `{n[:110]}` | {c} | {t / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * t / tot:.1f} |")
