# round 6 (second session), call 10: DevExec::wsum / wmax on DPP moves + v_readlane instead of six __shfl_xor steps (step kernel, aux row)
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_scale.py tests/test_gpu_marg_device.py tests/test_gpu_contract.py -x -q -m gpu 2>&1 | tail -4
for B in 8 64 512; do timeout 300 python tools/batch_profile.py $B 8 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_loop': [0-9.]*" | tr '\n' ' '; echo; done
} > $R/gpurun_out/r6b/call10.log 2>&1
cat $R/gpurun_out/r6b/call10.log
