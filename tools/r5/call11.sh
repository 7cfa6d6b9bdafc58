R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 300 python -m pytest tests/test_gpu_dev_solver.py tests/test_gpu_batch.py -x -q 2>&1 | tail -3
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mavx2 -I lio-mapping_amd/csrc tools/micro/pivot_chain.hip -o /tmp/pivot_chain && /tmp/pivot_chain | tail -1
for L in "" lio-mapping_amd/csrc/liblio_hip_b.so; do
  echo "== lib ${L:-default}"
  LIO_HIP_LIB=$L LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 timeout 200 python tools/batch_profile.py 64 2 2>&1 | grep "launch B" | tail -1
  LIO_HIP_LIB=$L timeout 200 python tools/batch_profile.py 64 6 2>&1 | tail -2 | cut -c1-420
done
