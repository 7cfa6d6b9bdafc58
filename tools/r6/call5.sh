# round 6, call 5: the control-block races of the step kernel fixed — identical windows at 64 / 160 / 512, batch and device-solver tests
mkdir -p gpurun_out/r6
{
python tools/r6/diag_determinism.py 64 24 1
python tools/r6/diag_determinism.py 160 6 1
python tools/r6/diag_determinism.py 512 3 1
python -m pytest tests/test_gpu_batch.py tests/test_gpu_dev_solver.py tests/test_gpu_marg_device.py -x -q 2>&1 | tail -5
} > gpurun_out/r6/call5.log 2>&1
grep "RESULT\|differ\|passed\|failed\|error" gpurun_out/r6/call5.log | cut -c1-300
