# round 6 (second session), call 9: the K-NN table in its rows form (entries for the occupied x-rows only) against the dense table
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q -m gpu 2>&1 | tail -6
for V in "LIO_BW_TABLE_ROWS=0" "LIO_BW_TABLE_ROWS=1"; do for B in 512 64 8; do echo "== $V B=$B"; env $V timeout 300 python tools/batch_profile.py $B 6 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_grid': [0-9.]*\|'dev_features': [0-9.]*\|'dev_rounds': [0-9.]*" | tr '\n' ' '; echo; done; done
} > $R/gpurun_out/r6b/call9.log 2>&1
cat $R/gpurun_out/r6b/call9.log
