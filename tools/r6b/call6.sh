# round 6 (second session), call 6: the aux row of the batched loop in its split form (csrc/aux_split.hip)
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q -m gpu 2>&1 | tail -5
for V in "LIO_BW_AUX_THREADS=64" ""; do for B in 512 256 128; do echo "== $V B=$B"; env $V timeout 300 python tools/batch_profile.py $B 6 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_loop': [0-9.]*" | tr '\n' ' '; echo; done; done
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b512 -o b -- python $R/tools/batch_profile.py 512 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_b512/b_results.db > $R/gpurun_out/r6b/aux_split_batch512_kernel_stats.md
head -14 $R/gpurun_out/r6b/aux_split_batch512_kernel_stats.md | cut -c1-180
} > $R/gpurun_out/r6b/call6.log 2>&1
cat $R/gpurun_out/r6b/call6.log
