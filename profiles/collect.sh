#!/bin/bash
# Collects the profiles committed under profiles/ (run on the GPU box from the repo root):
#   1. kernel trace + stats of the bench command the driver runs (c3, --steps 20 --warmup 5: 20 engine instances) -> <tag>_bench_default_{kernel_stats.csv,summary.txt}
#   2. kernel trace + stats of a single engine instance with its own (latency) tiles  -> <tag>_bench_w1_summary.txt (= roofline.isolated)
#   3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs)                              -> <tag>_c3_pmc_*.{csv,json}
#   4. PMC pass for MFMA utilisation (own run, kernel trace only for the durations)  -> <tag>_mfma_by_kernel.csv
#   5. phase table of the timed round (conv phases / coder-only / transitions)      -> <tag>_phase_table.txt
#   6. per-layer conv profile of one instance (tools/layer_profile.py)               -> <tag>_layer_profile_c3.txt
#   7. kernel stats of config 5 on one instance (tools/c5_profile.sh)                 -> <tag>_c5_stf_{4,1}x512x512_w1_summary.txt
# Counter passes run the eager launch path (RGBD_NO_GRAPH=1): every dispatch is then an ordinary kernel launch.
# Usage: bash profiles/collect.sh r04
set -e -o pipefail
tag=${1:-r04}
bench_args=${BENCH_ARGS:---steps 20 --warmup 5}  # the command line the round-end driver uses (BENCH_r01.json)
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" $bench_args --no-cpu-baseline --no-extras > "$out/stats.log" 2>&1
echo "[collect] default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_w1" -o run -- python3 "$root/bench.py" --workers 1 --steps 8 --warmup 4 --no-cpu-baseline --no-extras > "$out/stats_w1.log" 2>&1
echo "[collect] w1 done"
export RGBD_NO_GRAPH=1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_c3" -o run -- python3 "$root/bench.py" --steps 4 --warmup 4 --workers 1 --no-cpu-baseline --no-extras > "$out/fetch_c3.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write_c3" -o run -- python3 "$root/bench.py" --steps 4 --warmup 4 --workers 1 --no-cpu-baseline --no-extras > "$out/write_c3.log" 2>&1
echo "[collect] traffic done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/mfma" -o run -- python3 "$root/bench.py" --steps 4 --warmup 4 --workers 1 --no-cpu-baseline --no-extras > "$out/mfma.log" 2>&1
echo "[collect] mfma done"
unset RGBD_NO_GRAPH
cd "$root"
# every run executes warm-up steps, the timed steps and two conv-profile passes of two steps each: divide by all of them
steps=$(python3 -c "import json,sys;j=json.loads([l for l in open('$out/stats.log') if l.startswith('{')][-1]);print(j['steps']+j['config']['warmup_steps_run']+4*j['config'].get('steps_per_call',1))")
steps_w1=$(python3 -c "import json,sys;j=json.loads([l for l in open('$out/stats_w1.log') if l.startswith('{')][-1]);print(j['steps']+j['config']['warmup_steps_run']+4*j['config'].get('steps_per_call',1))")
# images-per-call x 512x640 padded pixels x 30.28 KB of layer-boundary bytes + 1.01 GB of weights per engine call,
# over the conv launches one call makes (round 5: a call codes four 4-image steps)
algo=$(python3 -c "import json,sys;j=json.loads([l for l in open('$out/stats.log') if l.startswith('{')][-1]);print(int((j['config'].get('images_per_call',4)*512*640*30.28e3+1.01e9)/j['roofline'].get('launches_per_call',j['roofline']['launches_per_step'])))")
imgs=$(python3 -c "import json,sys;j=json.loads([l for l in open('$out/stats.log') if l.startswith('{')][-1]);print(j['config'].get('images_per_call',4))")
{ python3 profiles/summarize.py "$out/stats" "$steps"; python3 profiles/timeline.py "$out/stats"; grep '^{' "$out/stats.log"; } > "$out/${tag}_bench_default_summary.txt"
{ python3 profiles/summarize.py "$out/stats_w1" "$steps_w1"; grep '^{' "$out/stats_w1.log"; } > "$out/${tag}_bench_w1_summary.txt"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_bench_default_kernel_stats.csv"
python3 profiles/pmc_traffic.py "$out/fetch_c3" "$out/write_c3" "$out/${tag}_c3" "$algo" "--workload c3_4x480x640, ${imgs} images per call (RGBD_NO_GRAPH=1)"
python3 profiles/mfma_util.py "$out/mfma" "$out/${tag}_mfma_by_kernel.csv"
python3 profiles/phase_table.py "$out/stats" "$out/stats.log" > "$out/${tag}_phase_table.txt"
LAYER_RAW="entropy_param|channel_context|local_context" python3 tools/layer_profile.py ${imgs} 512 640 > "$out/${tag}_layer_profile_c3.txt" 2>&1
# 7. config 5 (STF_united) on one engine instance, B = 4 and B = 1 -> <tag>_c5_stf_*_w1_summary.txt
bash tools/c5_profile.sh "$tag" > /dev/null 2>&1 || echo "[collect] c5 profile failed"
# the raw traces are large: keep the summaries
rm -rf "$out/stats" "$out/stats_w1" "$out/fetch_c3" "$out/write_c3" "$out/mfma"
ls "$out"
