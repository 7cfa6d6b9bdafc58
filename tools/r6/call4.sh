# round 6, call 4: is it co-residency on the CU?  The step workgroup with the whole LDS to itself; the Hessian of diverged windows
mkdir -p gpurun_out/r6
{
echo "=== LIO_BW_STEP_LDS_FULL"; LIO_BW_STEP_LDS_FULL=1 python tools/r6/diag_determinism.py 64 16 1
echo "=== baseline + Hessian check of the last trial"; LIO_DEBUG_DIGEST=1 python tools/r6/diag_determinism.py 64 6 1
} > gpurun_out/r6/call4.log 2>&1
grep "===\|RESULT\|differ" gpurun_out/r6/call4.log | cut -c1-300; grep "digest" gpurun_out/r6/call4.log | sort -k4 | awk '{print $4,$5,$6,$7,$8,$9,$10,$11,$12,$13}' | sort | uniq -c
