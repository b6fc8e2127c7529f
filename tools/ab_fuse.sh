#!/bin/bash
# same-box A/B of the fused launches: tools/ab_fuse.sh "VAR1 VAR2 ..."  -- each named switch set alone, against the default
# (RGBD_NO_FUSE, RGBD_NO_FUSE_LEAD, RGBD_NO_FUSE_WIDE, RGBD_NO_LEAD_SKIP), bench (K = 20) and isolated layer profile
mkdir -p gpurun_out/abf
vars=${1:-RGBD_NO_FUSE RGBD_NO_FUSE_LEAD}
for rep in 1 2; do for v in default $vars; do
  for u in $vars; do unset $u; done
  [ $v != default ] && export $v=1
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/abf/b_${v}_$rep.txt 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/abf/b_${v}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("$v rep $rep", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "iso", d["roofline"]["isolated"]["conv_ms_per_step"], d["roofline"]["isolated_timed_tiles"]["conv_ms_per_step"], flush=True)
PY
done; done
