# round 6, call 1: where do identical windows of a batch diverge?  (VERDICT r5 Weak #1)
mkdir -p gpurun_out/r6
{
python tools/r6/diag_determinism.py 64 6 1
python tools/r6/diag_determinism.py 64 4 3
LIO_BW_OCC=0 python tools/r6/diag_determinism.py 64 6 1
LIO_BW_GROUPS=1 python tools/r6/diag_determinism.py 64 6 1
python tools/r6/diag_determinism.py 8 4 1
python tools/r6/diag_determinism.py 160 3 1
} > gpurun_out/r6/call1.log 2>&1
tail -50 gpurun_out/r6/call1.log
