R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -5
for T in 0 64 128; do echo "LIO_BW_AUX_THREADS=$T"; LIO_BW_AUX_THREADS=$T timeout 200 python tools/batch_profile.py 512 8 2>&1 | tail -2 | cut -c1-100,180-420; done
timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-100,180-420
