# round 6 (second session), call 14: SQ counters of the batched moments kernel (tools/batched_moments.py 512: k_lidar_moments_batched, the body of k_bw_moments)
R=$PWD
mkdir -p $R/gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/prof_c$i -o c -- python $R/tools/batched_moments.py 512 > /dev/null 2>&1)
done
python - <<'PY' > $R/gpurun_out/r6b/moments_sq_counters.md
import sqlite3, glob
print("## SQ counters of k_lidar_moments_batched at 512 windows per launch (tools/batched_moments.py 512; one rocprofv3 --pmc pass per row group)\n")
print("| counter | launches | average per launch | avg duration us (profiled) |\n|---|---|---|---|")
for d in sorted(glob.glob("/tmp/prof_c*/*results.db")):
    cur = sqlite3.connect(d).cursor()
    try:
        q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name order by counter_name"
        for k, c, n, v, du in cur.execute(q):
            if "k_lidar_moments_batched" in k:
                print(f"| {c} | {n} | {v:.1f} | {(du or 0) / 1e3:.2f} |")
    except Exception as e:
        print("| (query failed:", e, ") | | | |")
PY
cat $R/gpurun_out/r6b/moments_sq_counters.md
