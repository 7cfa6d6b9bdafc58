R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_dev_solver.py tests/test_gpu_marg_device.py -x -q 2>&1 | tail -3
LIO_BW_AUX_STREAM=0 timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-420
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 LIO_BW_AUX_STREAM=0 timeout 200 python tools/batch_profile.py 64 2 2>&1 | grep "launch B" | tail -2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I lio-mapping_amd/csrc tools/micro/diag_block.hip -o /tmp/diag_block && timeout 60 /tmp/diag_block
