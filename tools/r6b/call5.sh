# round 6 (second session), call 5: k_odo_corr with four chunks of the ring-window scan in flight; finish threads A/B at 512 windows
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 900 python -m pytest tests/test_gpu_dropin_frontend.py tests/test_gpu_parity.py tests/test_gpu_ref_stages.py -x -q -m gpu -k "odom or Odom or frontend or dropin" 2>&1 | tail -4
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "lio-mapping_amd"))
import torch, bench
from lio_amd import capi
hip = capi.load_hip()
ds = bench.make_dataset("outdoor", 15)
for _ in range(3):
    print("point_odometry ms per scan (incl. h2d), packer mode:", bench.odometry_ms_per_scan(hip, ds))
PY
for V in "" "LIO_BW_FINISH_THREADS=8"; do echo "== $V"; env $V timeout 300 python tools/batch_profile.py 512 6 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'finish': [0-9.]*\|'describe': [0-9.]*" | tr '\n' ' '; echo; done
} > $R/gpurun_out/r6b/call5.log 2>&1
cat $R/gpurun_out/r6b/call5.log
