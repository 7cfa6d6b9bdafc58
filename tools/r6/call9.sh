# round 6, call 9: faster Jacobi (hardware reciprocal / rsqrt + Newton, division-free item map) — marg tests, phase clocks, batch points
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_marg_device.py tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q 2>&1 | tail -4
for B in 8 64 512; do LIO_DEBUG_DIGEST=1 python tools/batch_profile.py $B 6 2>&1 | grep -v "amdgpu.ids\|window [1-9]" | cut -c1-420; done
} > gpurun_out/r6/call9.log 2>&1
cat gpurun_out/r6/call9.log
