#!/bin/bash
# Round 4, GPU call 18: where the GPU idles inside one solve (kernel trace of the bench, gaps between kernels).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_g -o g -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --windows 0 --keyframes 0 --steps 10 --warmup 2 > /dev/null 2>&1)
python $R/profiles/gap_summary.py /tmp/prof_g/g_results.db > $O/gaps.md
cat $O/gaps.md
