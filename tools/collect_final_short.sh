# Round 3, last commit: the subset of tools/collect_profiles.sh whose numbers change with the host-side dogleg code and the
# resident kernel's partition / fold (the other kernels did not change, so the PMC passes of gpurun_out/r3final stand).
# gpurun --timeout 900 -- 'bash tools/collect_final_short.sh'
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3final3
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_dbg.json 2> $O/bench_dbg.err)
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
tail -3 $O/pytest_gpu.log
