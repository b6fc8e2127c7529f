#!/bin/bash
# same-box A/B of prebuilt libraries ab/<name>.so: single-instance and default benches (c2) + the c3 workload
set -e
mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "$@"; do
  export RGBD_AMD_LIB=$PWD/ab/$v.so
  timeout -k 10 200 python bench.py --workers 1 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/ab/lib_w1_${v}_$rep.txt 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab/lib_w16_${v}_$rep.txt 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload c3_4x480x640 > gpurun_out/ab/lib_c3_${v}_$rep.txt 2>&1
  python - <<PY
import json
for f in ("w1","w16","c3"):
    l=[x for x in open("gpurun_out/ab/lib_%s_${v}_$rep.txt"%f) if x.startswith("{")][-1]
    d=json.loads(l); print("$v rep $rep", f, d["ms_per_step"], d["value"], "conv iso ms", d["roofline"]["isolated"]["conv_ms_per_step"])
PY
done; done
