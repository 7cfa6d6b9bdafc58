# round 6, call 13: per-cell pruning in the one-lane-per-query K-NN walk — bit-identity across the lane counts, keyframe batch, timings
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_batch_scale.py tests/test_gpu_batch.py tests/test_gpu_kf_batch.py tests/test_knn_row_bound.py -x -q -m gpu 2>&1 | tail -6
for B in 64 512; do python tools/batch_profile.py $B 6 2>&1 | grep -v "amdgpu.ids" | cut -c1-420; done
python profiles/kf_batch_profile.py 2>&1 | tail -5 | cut -c1-300
} > gpurun_out/r6/call13.log 2>&1
cat gpurun_out/r6/call13.log
