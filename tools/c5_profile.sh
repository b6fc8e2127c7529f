#!/bin/bash
# Kernel stats of config 5 (STF_united, 512x512 RGB-D) on ONE engine instance: B = 4 and B = 1.   bash tools/c5_profile.sh <tag>
set -e -o pipefail
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for wl in c5_stf_4x512x512 c5_stf_1x512x512; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$wl" -o run -- python3 "$root/bench.py" --workload $wl --workers 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > "$out/stats_$wl.log" 2>&1
  steps=$(python3 -c "import json,sys;j=json.loads([l for l in open('$out/stats_$wl.log') if l.startswith('{')][-1]);print(j['steps']+j['config']['warmup_steps_run']+4)")
  { (cd "$root" && python3 profiles/summarize.py "$out/stats_$wl" "$steps"); grep '^{' "$out/stats_$wl.log"; } > "$out/${tag}_${wl}_w1_summary.txt"
  rm -rf "$out/stats_$wl"
done
ls "$out"
