R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for Q in 4 8; do for G in 1 2 4; do
  echo "== GPU_MAX_HW_QUEUES=$Q LIO_BW_GROUPS=$G"
  GPU_MAX_HW_QUEUES=$Q LIO_BW_GROUPS=$G timeout 200 python tools/batch_profile.py 64 4 2>&1 | tail -2 | cut -c1-400
done; done
