# Round 3, after the host-side dogleg changes: the subset of tools/collect_profiles.sh whose numbers changed (the kernels did not,
# so the PMC passes of gpurun_out/r3final stand).  gpurun --timeout 900 -- 'bash tools/collect_final_short.sh'
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3final2
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(LIO_RESIDENT_MOMENTS=0 LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_resident_off.json 2> $O/bench_resident_off.err)
(LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_dbg.json 2> $O/bench_dbg.err)
(LIO_HOST_SIGNAL=0 timeout 200 $B > $O/bench_no_host_signal.json 2> $O/bench_no_host_signal.err)
(timeout 200 python tools/stress_determinism.py 1000 > $O/stress.log 2>&1)
g++ -O3 -std=c++17 -ffp-contract=off -mavx2 -I lio-mapping_amd/csrc tools/micro/host_dogleg_pieces.cc -o /tmp/hdp && /tmp/hdp > $O/host_dogleg_pieces.txt 2>&1
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
tail -3 $O/pytest_gpu.log; tail -2 $O/stress.log; cat $O/host_dogleg_pieces.txt
