# round 6, call 10: triangle-of-blocks Jacobi — marg tests, phase clocks (profiling build), batch points
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_marg_device.py tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q 2>&1 | tail -4
LIO_HIP_LIB=lio-mapping_amd/csrc/liblio_hip_mprof.so LIO_DEBUG_DIGEST=1 python tools/batch_profile.py 64 3 2>&1 | grep "digest" | grep -v "window 63" | cut -c1-300
for B in 8 64 512; do LIO_DEBUG_DIGEST=1 python tools/batch_profile.py $B 6 2>&1 | grep -v "amdgpu.ids\|window [1-9]" | cut -c1-420; done
} > gpurun_out/r6/call10.log 2>&1
cat gpurun_out/r6/call10.log
