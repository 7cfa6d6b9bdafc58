# round 6 (second session), call 3: marginalization's eigensolves as Householder + implicit QL (tridiag_ql_lds) instead of cyclic Jacobi
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 900 python -m pytest tests/test_gpu_marg_device.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q -m gpu 2>&1 | tail -5
for B in 8 64 512; do LIO_DEBUG_DIGEST=1 timeout 300 python tools/batch_profile.py $B 8 2>&1 | grep -E "^B |digest\] window 0|dev_marg" | cut -c1-400; done
} > $R/gpurun_out/r6b/call3.log 2>&1
cat $R/gpurun_out/r6b/call3.log
