#!/bin/bash
# A/B of the admission cap of resident moments kernels with several windows in flight on one GPU (bench.py `batched` extra)
mkdir -p gpurun_out/cap
for wc in ${CAPS:-4:4 4:2 4:1 4:0 8:4 8:0}; do w=${wc%%:*}; cap=${wc##*:}; {
  LIO_MAX_RESIDENT_MOMENTS=$cap timeout 300 python bench.py --no-pmc --no-cpu-baseline --keyframes 0 --no-fed --windows $w --steps 50 > gpurun_out/cap/w${w}_cap$cap.json 2> gpurun_out/cap/w${w}_cap$cap.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/cap/w${w}_cap$cap.json").read().strip().splitlines()[-1])
print("windows $w cap $cap single", d["value"], "batched", d["batched"]["value"])
PY
}; done
