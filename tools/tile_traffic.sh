#!/bin/bash
# HBM fetch bytes and time of one conv shape under several forced tiles (rocprofv3 --pmc FETCH_SIZE; one run per tile).
# usage: bash tools/tile_traffic.sh "n cin h w cout k stride transposed" "wm,mt,nt,kc,dma" ...
shape=$1; shift
root=$(pwd); out=$root/gpurun_out/tt; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  tag=$(echo $t | tr ',' '_')
  RGBD_CONV_FORCE=$t timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$tag -o run -- python3 $root/tools/conv_one.py $shape > $out/$tag.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$out/$tag/**/*counter_collection.csv",recursive=True)
k=glob.glob("$out/$tag/**/*kernel_trace.csv",recursive=True)
tot=0;n=0
for r in csv.DictReader(open(f[0])):
    if 'conv_mfma' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        tot+=float(r['Counter_Value']); n+=1
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(k[0])) if 'conv_mfma' in r['Kernel_Name']]
print("$t", "launches",n, "FETCH_SIZE KiB/launch raw %.0f (x2 on gfx950 = %.1f MB)"%(tot/max(n,1), 2*tot/max(n,1)*1024/1e6), "min kernel us %.1f"%min(d))
PY
  rm -rf $out/$tag
done
