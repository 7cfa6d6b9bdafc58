#!/bin/bash
# Round 4, GPU call 8: k_ring_pick phase stamps on HDL-64E sweeps, sequential-speculative vs parallel-rounds picks; both against the reference's digests.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
for v in 0 1; do
  (LIO_PICK_ROUNDS=$v timeout 300 python -m pytest tests/test_gpu_ref_pointproc.py -q -x > $O/pytest_pp_$v.log 2>&1; echo rc=$? >> $O/pytest_pp_$v.log)
  grep -E "passed|failed|rc=" $O/pytest_pp_$v.log | tail -2
  (LIO_PICK_ROUNDS=$v LIO_DEBUG_TIMING=1 timeout 120 python profiles/pp_profile.py > $O/pp_$v.log 2>&1)
  grep -E "k_ring_pick ring|median" $O/pp_$v.log | tail -4
done
