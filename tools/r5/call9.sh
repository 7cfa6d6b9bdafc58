R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 400 python -m pytest tests/test_gpu_dev_solver.py tests/test_gpu_batch.py tests/test_gpu_marg_device.py -x -q 2>&1 | tail -4
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 64 2 2>&1 | grep "launch B" | tail -2
timeout 200 python tools/batch_profile.py 64 6 2>&1 | tail -2 | cut -c1-420
timeout 200 python tools/batch_profile.py 512 3 2>&1 | tail -2 | cut -c1-420
