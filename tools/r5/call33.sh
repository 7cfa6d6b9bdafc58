R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for T in 256 64 256 64; do echo "LIO_BW_AUX_THREADS=$T"; LIO_BW_AUX_THREADS=$T timeout 200 python tools/batch_profile.py 512 8 2>&1 | tail -2 | cut -c1-60,300-420; done
