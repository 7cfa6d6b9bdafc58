#!/bin/bash
# Round 4, GPU call 7: k_ring_pick in parallel rounds (bit-exact picks vs the reference's digests), its phase stamps; the opt-in one-launch VoxelGrid; suite.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
(LIO_DEBUG_TIMING=1 timeout 300 python -m pytest tests/test_gpu_ref_pointproc.py -q -s -x > $O/pytest_pp.log 2>&1; echo rc=$? >> $O/pytest_pp.log)
grep -E "pp timing|passed|failed|rc=" $O/pytest_pp.log | tail -8
(timeout 300 python -m pytest tests/test_gpu_vox_fused.py -q -s -x > $O/pytest_vox.log 2>&1; echo rc=$? >> $O/pytest_vox.log)
grep -E "per filter|passed|failed|rc=" $O/pytest_vox.log | tail -5
(timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pp -o pp -- python $R/profiles/pp_profile.py > $O/pp_profile.log 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_pp/pp_results.db > $O/pp_kernel_stats.md 2>/dev/null || true
head -12 $O/pp_kernel_stats.md
