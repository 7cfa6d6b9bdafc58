#!/bin/bash
# PointProcessor after the speculative pick: parity tests, per-kernel stats, wall time per sweep
O=gpurun_out/r3_pp; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -k "processor or rings or start_ori" 2>&1 | tail -3
python profiles/pp_profile.py | tail -1
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp_prof -o pp -- python $R/profiles/pp_profile.py > $R/$O/pp_profile.log 2>&1
cd $R; python profiles/summarize_rocpd.py /tmp/pp_prof/pp_results.db > $O/kernel_stats.md; head -20 $O/kernel_stats.md
