#!/bin/bash
# same-box A/B of prebuilt librgbd_amd.so variants under ab/: coder probe + single-instance bench, interleaved
set -e
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in "$@"; do
    RGBD_AMD_LIB=$PWD/ab/$v.so timeout -k 10 200 python tools/coder_probe.py > gpurun_out/ab/probe_${v}_$rep.txt 2>&1
    RGBD_AMD_LIB=$PWD/ab/$v.so timeout -k 10 200 python bench.py --workers 1 --steps 24 --warmup 6 > gpurun_out/ab/bench_${v}_$rep.txt 2>&1
    echo "$v $rep: $(tail -3 gpurun_out/ab/probe_${v}_$rep.txt | tr '\n' ' ' | cut -c1-300)"
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab/bench_${v}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("   bench ms_per_step", d["ms_per_step"])
PY
  done
done
