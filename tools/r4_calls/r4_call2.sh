#!/bin/bash
# Round 4, second GPU call: the host-side changes (split factorisation on by default, fused IMU blocks) on hardware.
#   full -m gpu suite; bench A/B (LIO_SPLIT_FACTOR 1 / 0) with the phase clocks; the store-acknowledgement micro-benchmark.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0"
run() { name=$1; shift; env "$@" LIO_DEBUG_TIMING=1 timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run split1 LIO_SPLIT_FACTOR=1
run split0 LIO_SPLIT_FACTOR=0
run split1b LIO_SPLIT_FACTOR=1
(timeout 300 python bench.py > $O/bench.json 2> $O/bench.err)
(hipcc --offload-arch=gfx950 -O3 -o /tmp/host_store_ack tools/micro/host_store_ack.hip && timeout 120 /tmp/host_store_ack > $O/host_store_ack.txt 2>&1)
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mavx2 -Wno-unused-function -I lio-mapping_amd/csrc tools/micro/host_dogleg_pieces.cc -x c++ -o /tmp/hdp 2>/dev/null || g++ -O3 -std=c++17 -ffp-contract=off -mavx2 -I lio-mapping_amd/csrc tools/micro/host_dogleg_pieces.cc -o /tmp/hdp; /tmp/hdp > $O/host_dogleg_pieces.txt 2>&1)
tail -3 $O/pytest_gpu.log
for f in $O/bench_*.json $O/bench.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
for n in split1 split0; do echo "== $n"; grep "dogleg\|resident\|evaluate" $O/bench_$n.err | tail -4; done
tail -30 $O/host_store_ack.txt
