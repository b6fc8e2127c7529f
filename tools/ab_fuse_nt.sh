#!/bin/bash
# same-box sweep of the fused tail's pixel-tile class: default plan vs RGBD_FUSE_NT=1/2/4 for every fusable site
mkdir -p gpurun_out/abf
for rep in 1 2; do for nt in plan 1 2 4; do
  if [ $nt = plan ]; then unset RGBD_FUSE_NT; else export RGBD_FUSE_NT=$nt; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 32 --warmup 8 > gpurun_out/abf/nt_${nt}_$rep.txt 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/abf/nt_${nt}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("FUSE_NT=$nt rep $rep", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "iso", d["roofline"]["isolated"]["conv_ms_per_step"], d["roofline"]["isolated_timed_tiles"]["conv_ms_per_step"], flush=True)
PY
done; done
