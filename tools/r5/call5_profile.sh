# kernel profile of the batched solve at B = 64 and B = 512 (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for B in 64 512; do
  (timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o b -- python $R/tools/batch_profile.py $B 3 > $O/r5_batch${B}_run.txt 2>&1)
  python $R/profiles/summarize_rocpd.py /tmp/prof_b$B/b_results.db > $O/r5_batch${B}_kernel_stats.md 2>&1
done
head -30 $O/r5_batch64_kernel_stats.md
tail -3 $O/r5_batch64_run.txt
