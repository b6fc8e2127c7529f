#!/bin/bash
# same-box A/B of the fused RB / ResidualUnit launches: nothing fused (RGBD_NO_FUSE), tails only (RGBD_NO_FUSE_LEAD), default
mkdir -p gpurun_out/abf
setv() { unset RGBD_NO_FUSE RGBD_NO_FUSE_LEAD; [ $1 = none ] && export RGBD_NO_FUSE=1; [ $1 = tail ] && export RGBD_NO_FUSE_LEAD=1; true; }
for rep in 1 2; do for v in none tail full; do
  setv $v
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 32 --warmup 8 > gpurun_out/abf/c3_${v}_$rep.txt 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/abf/c3_${v}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("fuse=$v rep $rep", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "iso", d["roofline"]["isolated"]["conv_ms_per_step"], d["roofline"]["isolated_timed_tiles"]["conv_ms_per_step"], flush=True)
PY
done; done
for v in none tail full; do
  setv $v
  timeout -k 10 200 python tools/layer_profile.py 4 512 640 > gpurun_out/abf/layers_${v}.txt 2>&1 || exit 1
  head -5 gpurun_out/abf/layers_${v}.txt | tail -4
done
