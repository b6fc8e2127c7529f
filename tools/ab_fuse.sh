#!/bin/bash
# same-box A/B of the fused RB / ResidualUnit tails: tools/ab_fuse.sh  (RGBD_NO_FUSE unset / set, interleaved)
mkdir -p gpurun_out/abf
for rep in 1 2; do for off in 0 1; do
  if [ $off = 1 ]; then export RGBD_NO_FUSE=1; else unset RGBD_NO_FUSE; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 32 --warmup 8 > gpurun_out/abf/c3_${off}_$rep.txt 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/abf/c3_${off}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("NO_FUSE=$off rep $rep", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "iso", d["roofline"]["isolated"]["conv_ms_per_step"], d["roofline"]["isolated_timed_tiles"]["conv_ms_per_step"], flush=True)
PY
done; done
for off in 0 1; do
  if [ $off = 1 ]; then export RGBD_NO_FUSE=1; else unset RGBD_NO_FUSE; fi
  timeout -k 10 200 python tools/layer_profile.py 4 512 640 > gpurun_out/abf/layers_${off}.txt 2>&1 || exit 1
  head -8 gpurun_out/abf/layers_${off}.txt
done
