# round 6 (second session), call 11: the loop groups' chains enqueued interleaved (iteration by iteration) instead of one whole chain after the other; 2 / 3 groups
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q -m gpu 2>&1 | tail -3
for V in "" "LIO_BW_GROUPS=3" "LIO_BW_GROUPS=4"; do for B in 8 32 64 128 512; do echo "== $V B=$B"; env $V timeout 300 python tools/batch_profile.py $B 8 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_loop': [0-9.]*\|'solve': [0-9.]*" | tr '\n' ' '; echo; done; done
} > $R/gpurun_out/r6b/call11.log 2>&1
cat $R/gpurun_out/r6b/call11.log
