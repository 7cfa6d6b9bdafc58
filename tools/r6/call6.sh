# round 6, call 6: the new at-scale tests + the batch tests
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_batch_scale.py tests/test_gpu_batch.py tests/test_gpu_dev_solver.py -x -q -s 2>&1 | tail -40 > gpurun_out/r6/call6.log
cat gpurun_out/r6/call6.log
