#!/bin/bash
# forced tile variants on the large-layer shapes (RGBD_CONV_FORCE = wm,mt,nt,kc,dma; empty = table / cost model)
for f in "" "2,3,8,16,1" "2,2,8,16,1" "2,3,8,16,0" "2,3,8,64,0"; do for s in "8 96 128 128 96 3 1 0" "8 384 128 128 192 3 1 0" "8 192 128 128 96 1 1 0" "8 96 128 128 192 1 1 0" "8 384 128 128 192 5 2 0" "8 192 64 64 192 5 2 1" "8 96 64 64 96 3 1 0" "4 96 256 320 96 3 1 0"; do
  echo -n "force=[$f] $s : "; RGBD_CONV_FORCE=$f timeout -k 5 60 python tools/conv_one.py $s 2>/dev/null | tail -1 | awk '{print $4, $NF}'
done; done
