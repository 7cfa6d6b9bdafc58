#!/usr/bin/env python
"""Idle time between the kernels of one solve, from a rocprofv3 (rocpd sqlite) kernel trace of bench.py.
Usage: gap_summary.py results.db > gaps.md
Takes a solve from the MIDDLE of the trace (from one k_transform_concat to the next), lists every kernel with its start offset, duration
and the gap to the end of whatever ended last before it (all streams), and sums busy / idle time."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_transform_concat" in r[0]]
if len(starts) < 3:
    print("not enough solves in the trace")
    sys.exit(0)
mid = len(starts) // 2
a, b = starts[mid], starts[mid + 1]
seg = rows[a:b]
t0 = seg[0][1]
print("| # | kernel | start us | duration us | gap before us |")
print("|---|---|---|---|---|")
last_end = t0
busy = 0.0
idle = 0.0
for i, (n, s, e) in enumerate(seg):
    gap = (s - last_end) / 1e3
    short = n.split("(")[0].replace("void ", "")[-60:]
    print(f"| {i} | `{short}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} |")
    if gap > 0:
        idle += gap
    busy += (min(e, seg[i + 1][1]) - s) / 1e3 if i + 1 < len(seg) and seg[i + 1][1] < e else (e - s) / 1e3
    last_end = max(last_end, e)
print()
print(f"solve span {(rows[b][1] - t0) / 1e3:.1f} us, {len(seg)} kernels, idle between kernels {idle:.1f} us")
