#!/bin/bash
# kernel timeline of the pipelined harness (rocprofv3 --kernel-trace of tools/harness_throughput.py): GPU occupancy and the kernels' shares
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && export HARNESS_NO_PNG=1 HARNESS_STAGES=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/hprof -o run -- python3 $root/tools/harness_throughput.py ${1:-96} ${2:-8:4} > $root/gpurun_out/hprof.log 2>&1
cd $root
grep "workers\|host seconds" gpurun_out/hprof.log
python3 profiles/timeline.py gpurun_out/hprof
python3 - <<EOF
import csv,glob
from collections import defaultdict
rows=list(csv.DictReader(open(glob.glob("gpurun_out/hprof/**/*kernel_trace.csv",recursive=True)[0])))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
t0=ev[len(ev)//2][0]
d=defaultdict(float)
for s,e,n in ev:
    if s>=t0: d[n.split("(")[0][:60]]+=e-s
tot=sum(d.values())
for k,v in sorted(d.items(),key=lambda kv:-kv[1])[:14]: print(f"{k:62s} {v/1e6:9.1f} ms {100*v/tot:5.1f}%")
EOF
rm -rf gpurun_out/hprof
