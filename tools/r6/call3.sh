# round 6, call 3: A/B of the diverging batched loop — baseline, full barriers, padded per-window records, both groups on one stream
mkdir -p gpurun_out/r6
V=lio-mapping_amd/csrc
{
echo "=== baseline"; python tools/r6/diag_determinism.py 64 12 1
echo "=== LIO_SYNC_FULL"; LIO_HIP_LIB=$V/liblio_hip_sync.so python tools/r6/diag_determinism.py 64 12 1
echo "=== LIO_PAD_WINDOWS"; LIO_HIP_LIB=$V/liblio_hip_pad.so python tools/r6/diag_determinism.py 64 12 1
echo "=== LIO_BW_GROUPS_SERIAL"; LIO_BW_GROUPS_SERIAL=1 python tools/r6/diag_determinism.py 64 12 1
} > gpurun_out/r6/call3.log 2>&1
grep -c "differ" gpurun_out/r6/call3.log; grep "===\|RESULT\|differ" gpurun_out/r6/call3.log | cut -c1-400
