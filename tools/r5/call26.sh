R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for G in 1 4; do echo "B=512 LIO_BW_GROUPS=$G"; LIO_BW_GROUPS=$G timeout 200 python tools/batch_profile.py 512 6 2>&1 | tail -2 | cut -c1-100,180-420; done
echo "B=64 LIO_BW_LPQ=2"; LIO_BW_LPQ=2 timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-100,180-420
echo "B=64 LIO_BW_GROUPS=3"; LIO_BW_GROUPS=3 timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-100,180-420
