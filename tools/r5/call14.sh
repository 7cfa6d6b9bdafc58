R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -3
echo "== aux side stream (default) vs in-line, G=2, 4 HW queues"
for A in 1 0; do echo "LIO_BW_AUX_STREAM=$A"; LIO_BW_AUX_STREAM=$A timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-420; done
echo "== GPU_MAX_HW_QUEUES=8"
for G in 2 4; do echo "G=$G"; GPU_MAX_HW_QUEUES=8 LIO_BW_GROUPS=$G timeout 200 python tools/batch_profile.py 64 8 2>&1 | tail -2 | cut -c1-420; done
echo "== B=512 / B=8, default"
for B in 512 8; do timeout 200 python tools/batch_profile.py $B 5 2>&1 | tail -2 | cut -c1-420; done
GPU_MAX_HW_QUEUES=8 LIO_BW_GROUPS=4 timeout 200 python tools/batch_profile.py 512 5 2>&1 | tail -2 | cut -c1-420
echo "== stamps of launch B (G=1, aux in line)"
LIO_DEBUG_TIMING=1 LIO_BW_GROUPS=1 LIO_BW_AUX_STREAM=0 timeout 200 python tools/batch_profile.py 64 2 2>&1 | grep "launch B" | tail -2
