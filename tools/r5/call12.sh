R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 64 2 2>&1 | grep "launch B" | tail -1
for B in 64 512 8; do timeout 200 python tools/batch_profile.py $B 5 2>&1 | tail -2 | cut -c1-420; done
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -o b -- python $R/tools/batch_profile.py 64 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_b64/b_results.db > $O/r5_c_batch64_kernel_stats.md 2>&1
head -14 $O/r5_c_batch64_kernel_stats.md | cut -c1-150
