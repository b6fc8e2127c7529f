#!/bin/bash
# Collects the profiles committed under profiles/ (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command       -> <tag>_bench_default_{kernel_stats.csv,summary.txt}
#   2. kernel trace + stats of a single engine instance running the default run's (throughput) tiles
#                                                                -> <tag>_bench_w1_summary.txt   (matches roofline.isolated)
#   3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, no trace domains) -> <tag>_pmc_*.{csv,json}
# Usage: bash profiles/collect.sh r01
set -e -o pipefail
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" --no-cpu-baseline > "$out/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_w1" -o run -- python3 "$root/bench.py" --workers 1 --tile-mode throughput --steps 6 --warmup 2 --no-cpu-baseline > "$out/stats_w1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o run -- python3 "$root/bench.py" --steps 2 --warmup 1 --workers 1 --tile-mode throughput --no-cpu-baseline > "$out/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o run -- python3 "$root/bench.py" --steps 2 --warmup 1 --workers 1 --tile-mode throughput --no-cpu-baseline > "$out/write.log" 2>&1
cd "$root"
# every run executes warm-up steps, the timed steps and two single-instance steps (roofline.isolated): divide by all of them
steps=$(python3 -c "import json,sys;j=json.loads([l for l in open('$out/stats.log') if l.startswith('{')][-1]);print(j['steps']+max(j['warmup'],min(j['config']['engine_instances'],j['steps']))+2)")
{ python3 profiles/summarize.py "$out/stats" "$steps"; python3 profiles/timeline.py "$out/stats"; grep '^{' "$out/stats.log"; } > "$out/${tag}_bench_default_summary.txt"
{ python3 profiles/summarize.py "$out/stats_w1" 10; grep '^{' "$out/stats_w1.log"; } > "$out/${tag}_bench_w1_summary.txt"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_bench_default_kernel_stats.csv"
python3 profiles/pmc_traffic.py "$out/fetch" "$out/write" "$out/$tag" 28420875.4
# the same two PMC passes for the 480x640 workload (BASELINE config 3's per-GPU share): 4 x 512x640 padded pixels x 30.28 KB
# + 1.01 GB weights per step over 594 launches = 68.5 MB algorithmic per launch
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_c3" -o run -- python3 "$root/bench.py" --workload c3_4x480x640 --steps 2 --warmup 1 --workers 1 --tile-mode throughput --no-cpu-baseline > "$out/fetch_c3.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write_c3" -o run -- python3 "$root/bench.py" --workload c3_4x480x640 --steps 2 --warmup 1 --workers 1 --tile-mode throughput --no-cpu-baseline > "$out/write_c3.log" 2>&1
cd "$root"
python3 profiles/pmc_traffic.py "$out/fetch_c3" "$out/write_c3" "$out/${tag}_c3" 68516164 "--workload c3_4x480x640"
ls "$out"
