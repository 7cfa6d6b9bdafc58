R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 64 512; do
  (timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o b -- python $R/tools/batch_profile.py $B 3 > /dev/null 2>&1)
  python $R/profiles/summarize_rocpd.py /tmp/prof_b$B/b_results.db > $O/batch${B}_kernel_stats.md
  head -14 $O/batch${B}_kernel_stats.md | cut -c1-150
done
