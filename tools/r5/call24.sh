R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for S in 4096 8192 16384; do echo "LIO_BW_SLOTS_PER_BLOCK=$S"; for B in 64 512; do LIO_BW_SLOTS_PER_BLOCK=$S timeout 200 python tools/batch_profile.py $B 8 2>&1 | tail -2 | cut -c1-100,180-420; done; done
