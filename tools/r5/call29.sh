R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed"
for B in 512 64 512; do timeout 200 python tools/batch_profile.py $B 8 2>&1 | tail -2 | cut -c1-100,180-420; done
