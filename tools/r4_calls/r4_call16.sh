#!/bin/bash
# Round 4, GPU call 16: per-block stamps of k_vox_fused in the bench's steady state.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(LIO_VOX_FUSED=1 LIO_DEBUG_TIMING=1 timeout 300 $B --steps 20 > $O/bench_fused1_dbg.json 2> $O/bench_fused1_dbg.err)
grep "k_vox_fused\|done:" $O/bench_fused1_dbg.err | tail -8
