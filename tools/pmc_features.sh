# SQ counters of the search kernels (one rocprofv3 --pmc pass per counter group, kernel trace only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pmcf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
run() { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/p_$tag -o p -- python $R/bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 1 --windows 0 --keyframes 0 > /dev/null 2> $O/err_$tag.txt; python - "$tag" <<'PY'
import sqlite3, sys, os, glob
tag = sys.argv[1]
db = glob.glob(f"/tmp/p_{tag}/*results.db")
out = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r2pmcf", f"pmc_{tag}.md"), "w")
if not db:
    out.write("no db\n"); sys.exit(0)
cur = sqlite3.connect(db[0]).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, n, v in rows:
    if "k_odom_round" in k or "k_features" in k or "k_odom_update" in k or "k_lidar_moments" in k:
        out.write(f"{k[:60]} | {c} | {n} | {v:.1f}\n")
PY
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
run c SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run d GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
cat $O/pmc_*.md
