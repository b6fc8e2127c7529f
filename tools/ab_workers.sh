#!/bin/bash
# same-box sweep of the number of engine instances per GPU (bench.py --workers; CodecPool refuses more than 32)
mkdir -p gpurun_out/abw
for rep in 1 2; do for w in ${WORKERS:-12 16 20 24 32}; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --workers $w ${BENCH_ARGS} > gpurun_out/abw/w${w}_$rep.txt 2>&1 || { tail -5 gpurun_out/abw/w${w}_$rep.txt; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/abw/w${w}_$rep.txt") if x.startswith("{")][-1]
d=json.loads(l); print("workers=$w rep $rep", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "host cores", d["config"]["host_cores_busy_per_rank"], flush=True)
PY
done; done
