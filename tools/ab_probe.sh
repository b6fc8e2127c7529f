#!/bin/bash
# same-box A/B of the coder probe (ns per symbol by class of CDF rows, rocprofv3 kernel trace) for prebuilt libraries ab/<name>.so
root=$(pwd)
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  export RGBD_AMD_LIB=$root/ab/$v.so
  rm -rf $root/gpurun_out/ab/probe_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/ab/probe_$v -o run -- python3 $root/tools/coder_probe.py 200000 > $root/gpurun_out/ab/probe_$v.log 2>&1
  echo "== $v rep $rep"; python3 $root/tools/coder_probe_report.py $root/gpurun_out/ab/probe_$v 200000
  rm -rf $root/gpurun_out/ab/probe_$v
done; done
