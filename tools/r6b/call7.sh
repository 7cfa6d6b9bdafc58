# round 6 (second session), call 7: identical windows of a batch at the final commit — every stage's digests over repeated trials (tools/r6/diag_determinism.py)
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 600 python tools/r6/diag_determinism.py 64 24 1 2>&1 | tail -4
timeout 600 python tools/r6/diag_determinism.py 64 6 3 2>&1 | tail -3
timeout 600 python tools/r6/diag_determinism.py 160 6 1 2>&1 | tail -3
timeout 900 python tools/r6/diag_determinism.py 512 4 1 2>&1 | tail -3
timeout 600 python tools/r6/diag_determinism.py 8 8 2 2>&1 | tail -3
timeout 600 python tools/stress_determinism.py 200 2>&1 | tail -4
} > $R/gpurun_out/r6b/call7.log 2>&1
cat $R/gpurun_out/r6b/call7.log
