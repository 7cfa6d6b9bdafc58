R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_dropin_frontend.py tests/test_gpu_ref_pointproc.py tests/test_gpu_dropin.py tests/test_abi.py tests/test_gpu_rccl.py "tests/test_gpu_parity.py" -x -q -s 2>&1 | grep -E "passed|failed|Error|error|PointProcessorHip|PointOdometryHip|free|assert" | tail -40
