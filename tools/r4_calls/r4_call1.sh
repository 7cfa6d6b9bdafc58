# Round 4, first GPU call of this session: the whole -m gpu suite (every failure listed), then the default bench line.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 300 python bench.py > $O/bench.json 2> $O/bench.err)
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -30; cut -c1-600 $O/bench.json
