# round 6, call 11: segmented sort in the batched filter and K-NN grid (no rocPRIM in the chain) — batch tests, then the batch points
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_seg_sort.py tests/test_gpu_batch.py tests/test_gpu_batch_scale.py tests/test_gpu_dev_solver.py -x -q -m gpu 2>&1 | tail -6
for B in 8 64 512; do python tools/batch_profile.py $B 6 2>&1 | grep -v "amdgpu.ids" | cut -c1-420; done
} > gpurun_out/r6/call11.log 2>&1
cat gpurun_out/r6/call11.log
