R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_dev_solver.py -x -q 2>&1 | tail -2
for S in 0 1024 4096; do echo "LIO_BW_SLOTS_PER_BLOCK=$S"; LIO_BW_SLOTS_PER_BLOCK=$S timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-420; done
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 64 1 2>&1 | grep "launch B\|aux row" | tail -3 | head -1
