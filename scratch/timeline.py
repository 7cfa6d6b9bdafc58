import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in cols if c in ("stream_id", "queue_id", "stream", "queue")]
rows = cur.execute(f"select {name_col}, start, end {''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_line_features" in r[0]]
i0 = idx[-3]
t0 = rows[i0][1]
for r in rows[i0 - 2:i0 + 14]:
    print("%-40s start %8.1f us  dur %7.1f us %s" % (r[0][:40], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3:]))
