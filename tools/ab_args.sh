#!/bin/bash
# same-box comparison of bench.py argument / environment variants (interleaved, two repetitions):
#   bash tools/ab_args.sh "name1|ENV=.. --args" "name2|--args" ...
mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "$@"; do
  name=${v%%|*}; rest=${v#*|}
  envs=""; args=""
  for tok in $rest; do case $tok in *=*) if [[ $tok != --* ]]; then envs="$envs $tok"; else args="$args $tok"; fi;; *) args="$args $tok";; esac; done
  env $envs timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-extras > gpurun_out/ab/args_${name}_$rep.txt 2>&1
  python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/ab/args_${name}_$rep.txt") if x.startswith("{")][-1]
    d=json.loads(l); print("$name rep $rep: ms/step", d["ms_per_step"], "Mpx/s", d["value"], "frac", d["roofline"]["frac"], "inst", d["config"]["engine_instances"], "x", d["config"].get("images_per_call"), flush=True)
except Exception as e:
    print("$name rep $rep: failed", e, flush=True)
PY
done; done
