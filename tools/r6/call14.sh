# round 6, call 14: keyframe batch with the queries processed in map-cell order
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_kf_batch.py tests/test_gpu_mapping.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
python profiles/kf_batch_profile.py 2>&1 | tail -3 | cut -c1-400
LIO_KF_LPQ=2 python profiles/kf_batch_profile.py 2>&1 | tail -1 | cut -c1-300
} > gpurun_out/r6/call14.log 2>&1
cat gpurun_out/r6/call14.log
