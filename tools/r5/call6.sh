R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_dev_solver.py -x -q 2>&1 | tail -5 > $O/r5_call6_tests.log
cat $O/r5_call6_tests.log
for L in 0 4 2; do
  echo "== LIO_BW_LPQ=$L"
  LIO_BW_LPQ=$L timeout 200 python tools/batch_profile.py 64 4 2>&1 | tail -2
done
LIO_BW_LPQ=0 timeout 200 python tools/batch_profile.py 512 3 2>&1 | tail -2
LIO_BW_LPQ=0 timeout 200 python tools/batch_profile.py 8 6 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -o b -- python $R/tools/batch_profile.py 64 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_b64/b_results.db > $O/r5_b_batch64_kernel_stats.md 2>&1
head -24 $O/r5_b_batch64_kernel_stats.md | cut -c1-150
