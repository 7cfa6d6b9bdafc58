R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for O in 0 6 8; do echo "LIO_BW_OCC=$O"; for B in 64 512; do LIO_BW_OCC=$O timeout 200 python tools/batch_profile.py $B 8 2>&1 | tail -2 | cut -c1-100,180-420; done; done
LIO_BW_LPQ=1 LIO_BW_OCC=8 timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed"
