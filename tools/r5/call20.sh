R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_dev_solver.py tests/test_gpu_marg_device.py tests/test_gpu_kf_batch.py -x -q 2>&1 | tail -2
for B in 64 512 8; do timeout 200 python tools/batch_profile.py $B 6 2>&1 | tail -2 | cut -c1-420; done
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 64 1 2>&1 | grep "launch B\|aux row" | tail -3
timeout 400 python bench.py --workload keyframes --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-900
