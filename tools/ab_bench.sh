#!/bin/bash
# same-box A/B of prebuilt libraries ab/<name>.so on the driver's bench command (interleaved, two repetitions):
#   bash tools/ab_bench.sh "<bench args>" name1 name2 ...
set -e
args=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "$@"; do
  RGBD_AMD_LIB=$PWD/ab/$v.so timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-extras > gpurun_out/ab/bench_${v}_$rep.txt 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/ab/bench_${v}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("$v rep $rep: ms/step", d["ms_per_step"], "Mpx/s", d["value"], "frac", d["roofline"]["frac"], "inst", d["config"]["engine_instances"], "x", d["config"].get("images_per_call"), "iso(timed tiles) ms", d["roofline"]["isolated_timed_tiles"]["conv_ms_per_step"], flush=True)
PY
done; done
