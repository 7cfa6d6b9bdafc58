# round 6, call 8: kernel trace of the batched step at 64 windows after the Jacobi change
mkdir -p gpurun_out/r6
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -o b -- python $R/tools/batch_profile.py 64 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_b64/b_results.db > $R/gpurun_out/r6/call8_batch64_kernel_stats.md
head -30 $R/gpurun_out/r6/call8_batch64_kernel_stats.md | cut -c1-200
