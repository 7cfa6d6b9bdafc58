# round 6, call 16: segmented sort with the scatter staged through LDS — sort test, batch tests, A/B of tile sizes / staging at 64 and 512 windows
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_seg_sort.py tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q -m gpu 2>&1 | tail -4
for V in "" "LIO_SS_UNSTAGED=1" "LIO_SS_THREADS=256" "LIO_SS_THREADS=1024"; do
  for B in 64 512; do echo "== $V B=$B"; env $V python tools/batch_profile.py $B 6 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_filter': [0-9.]*\|'dev_grid': [0-9.]*" | tr '\n' ' '; echo; done
done
} > gpurun_out/r6/call16.log 2>&1
cat gpurun_out/r6/call16.log
