#!/usr/bin/env python
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; ROCm 7.2 rocpd
sqlite output).  Usage: pmc_summary.py fetch.db write.db out.json > table.md
Units: both counters are in KB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half
the bytes of a wide (16 B/lane) coalesced streaming read -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 for
kernels whose reads are float4 streams (all of ours); WRITE_SIZE is uncalibrated on gfx950."""
import json
import sqlite3
import sys


def load(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = "select kernel_name, grid_size, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name, grid_size"
    return {(k, g): (n, v) for k, g, n, v in cur.execute(q, (counter,))}


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
rows = []
for key in sorted(set(f) | set(w), key=lambda kg: -(f.get(kg, (0, 0))[0] * f.get(kg, (0, 0))[1])):
    k, g = key
    if "lio::" not in k.split("(")[0]:
        continue
    nf, vf = f.get(key, (0, 0.0))
    nw, vw = w.get(key, (0, 0.0))
    short = k.split("(")[0].replace("void ", "")
    hbm = (2 * vf + vw) * 1024
    out.setdefault(short, []).append({"grid_size": g, "launches": nf, "fetch_kb": vf, "write_kb": vw, "hbm_bytes_corrected": hbm})
    rows.append((short, g, nf, vf, vw, hbm))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("| kernel | grid threads | launches | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch | HBM bytes/launch (2*F+W)*1024 |")
print("|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r[0]}` | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.0f} |")
