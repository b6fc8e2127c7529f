#!/bin/bash
# same-box A/B at the default bench configuration (W=16) and c3
set -e
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in "$@"; do
    RGBD_AMD_LIB=$PWD/ab/$v.so timeout -k 10 300 python bench.py > gpurun_out/ab/bench16_${v}_$rep.txt 2>&1
    RGBD_AMD_LIB=$PWD/ab/$v.so timeout -k 10 300 python bench.py --workload c3_4x480x640 --steps 24 --warmup 8 > gpurun_out/ab/benchc3_${v}_$rep.txt 2>&1
    python - <<PY
import json
for f in ("bench16","benchc3"):
    l=[x for x in open("gpurun_out/ab/%s_${v}_$rep.txt"%f) if x.startswith("{")][-1]
    d=json.loads(l); print("$v $rep", f, d["ms_per_step"], d["value"])
PY
  done
done
