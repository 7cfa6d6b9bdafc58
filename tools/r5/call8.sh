R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -3
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 64 2 2>&1 | grep "launch B\|solves/s" | tail -3
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 1 2 2>&1 | grep "launch B\|solves/s" | tail -3
timeout 200 python tools/batch_profile.py 64 6 2>&1 | tail -2 | cut -c1-420
timeout 200 python tools/batch_profile.py 8 6 2>&1 | tail -2 | cut -c1-420
