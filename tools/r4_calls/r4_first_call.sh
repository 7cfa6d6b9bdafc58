# Round 4, first GPU call: what round 3 wrote after its GPU budget was spent and could only dry-run on the CPU.
#   gpurun --timeout 1500 -- 'bash tools/r4_first_call.sh'        (writes gpurun_out/r4a/)
# 1. the six GPU tests against the reference-produced golden files (tests/test_gpu_ref_estimator.py, tests/test_gpu_ref_stages.py) — alone
#    first, with output, so that their measured gaps are on record even if something else in the suite fails;
# 2. the whole -m gpu suite;
# 3. the product against the reference's Estimator.cc step by step (teacher-forced; tools/gpu_ref_estimator_gaps.py) — the numbers
#    that turn it into a test (read its docstring for what the first W steps after the initialisation will show);
# 4. the default bench line, as a check that nothing moved;
# 5. tools/micro/host_store_ack.hip: what publishing a frame record to the host costs a wave (first bullet).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
(timeout 400 python -m pytest tests/test_gpu_zz_ref_state.py tests/test_gpu_ref_estimator.py tests/test_gpu_ref_stages.py tests/test_gpu_ref_pointproc.py -q -s -m gpu > $O/pytest_gpu_ref.log 2>&1; echo rc=$? >> $O/pytest_gpu_ref.log)
(timeout 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 600 python tools/gpu_ref_estimator_gaps.py indoor indoor_12_7 outdoor64 > $O/ref_estimator_gaps.txt 2> $O/ref_estimator_gaps.err)
(timeout 400 python bench.py > $O/bench.json 2> $O/bench.err)
(hipcc --offload-arch=gfx950 -O3 -o /tmp/host_store_ack tools/micro/host_store_ack.hip && timeout 120 /tmp/host_store_ack > $O/host_store_ack.txt 2>&1)
tail -5 $O/pytest_gpu_ref.log $O/pytest_gpu.log; tail -3 $O/ref_estimator_gaps.txt; cat $O/bench.json | cut -c1-400
