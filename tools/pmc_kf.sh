R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/p_$tag -o p -- python $R/bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 1 --windows 0 --keyframes 300 > /dev/null 2> $O/err_$tag.txt; python - "$tag" <<'PY'
import sqlite3, sys, os, glob
tag = sys.argv[1]
db = glob.glob(f"/tmp/p_{tag}/*results.db")
out = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r3e", f"pmc_{tag}.md"), "w")
cur = sqlite3.connect(db[0]).cursor()
for k, c, n, v, mx in cur.execute("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection group by kernel_name, counter_name"):
    if "k_kf_" in k:
        out.write(f"{k[:40]} | {c} | {n} | avg {v:.1f} | max {mx:.1f}\n")
PY
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
run b SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
cat $O/pmc_*.md
