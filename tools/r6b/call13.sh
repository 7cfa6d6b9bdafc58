# round 6 (second session), call 13: a batch of >= 128 windows solved as two parts side by side from two host threads (capi.hip: batch_for_parts)
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 1800 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_scale.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -4
for V in "LIO_BW_PARTS=1" ""; do for B in 64 128 256 512; do echo "== $V B=$B"; env $V timeout 300 python tools/batch_profile.py $B 8 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_loop': [0-9.]*\|'total': [0-9.]*" | tr '\n' ' '; echo; done; done
LIO_BW_PARTS=2 timeout 300 python tools/batch_profile.py 64 8 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s"
} > $R/gpurun_out/r6b/call13.log 2>&1
cat $R/gpurun_out/r6b/call13.log
