python - <<'PY'
import pickle, sys, os
sys.path.insert(0, 'lio-mapping_amd'); sys.path.insert(0, '.')
import torch
import bench
from lio_amd import capi
hip = capi.load_hip()
ds = bench.make_dataset("outdoor", 15, 0.0)
clouds, _ = bench.feature_clouds(hip, ds)
pickle.dump((ds, clouds), open('/tmp/w.pkl', 'wb'))
PY
python bench.py --pmc-child /tmp/w.pkl --steps 3 --workload hdl64; echo child rc=$?
cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d /tmp/pq -o pq -- python $OLDPWD/bench.py --pmc-child /tmp/w.pkl --steps 3 --workload hdl64 > /tmp/pq.log 2>&1; echo prof rc=$?
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/pq/**/*.db', recursive=True)
print(db)
cur = sqlite3.connect(db[0]).cursor()
for k, g, n, v, d in cur.execute("select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection where counter_name='SQ_INSTS_VALU' group by kernel_name, grid_size order by count(*) desc limit 40"):
    print(k[:60], g, n, round(v), round(d))
PY
