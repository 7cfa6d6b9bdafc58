# round 6 (second session), call 2: PointProcessor as one launch chain over B sweeps — parity tests, throughput, kernel profile
mkdir -p gpurun_out/r6b
R=$PWD
{
timeout 900 python -m pytest tests/test_gpu_pp_batch.py tests/test_gpu_ref_pointproc.py tests/test_gpu_dropin_frontend.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "point_processor or pp or start_ori or ring" 2>&1 | tail -5
for B in 1 8 64 256; do timeout 200 python tools/pp_batch_profile.py $B 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pp -o pp -- python $R/tools/pp_batch_profile.py 64 8 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_pp/pp_results.db > $R/gpurun_out/r6b/pp_batch64_kernel_stats.md
head -16 $R/gpurun_out/r6b/pp_batch64_kernel_stats.md | cut -c1-200
} > gpurun_out/r6b/call2.log 2>&1
cat $R/gpurun_out/r6b/call2.log
