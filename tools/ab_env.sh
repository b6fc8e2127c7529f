#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh VAR  (runs the benches with VAR unset and VAR=1, interleaved)
set -e
mkdir -p gpurun_out/ab
v=$1
for rep in 1 2; do for on in 0 1; do
  if [ $on = 1 ]; then export $v=1; else unset $v; fi
  timeout -k 10 200 python bench.py --workers 1 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/ab/env_w1_${on}_$rep.txt 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab/env_w16_${on}_$rep.txt 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload c3_4x480x640 --steps 24 --warmup 8 > gpurun_out/ab/env_c3_${on}_$rep.txt 2>&1
  python - <<PY
import json
for f in ("w1","w16","c3"):
    l=[x for x in open("gpurun_out/ab/env_%s_${on}_$rep.txt"%f) if x.startswith("{")][-1]
    d=json.loads(l); print("$v=$on rep $rep", f, d["ms_per_step"], d["value"], "conv iso ms", d["roofline"]["isolated"]["conv_ms_per_step"])
PY
done; done
