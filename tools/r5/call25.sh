R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for B in 64 512 8; do timeout 200 python tools/batch_profile.py $B 8 2>&1 | tail -2 | cut -c1-420; done
