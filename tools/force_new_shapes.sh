#!/bin/bash
# forced tile variants on the shapes of the packed image layers (not in the tile tables): K-packed first conv as 1x1
# (80 / 32 -> 192 at 256x320) and the sub-pixel last deconv as 3x3 (192 -> 16 at 256x320); RGBD_CONV_FORCE = wm,mt,nt,kc,dma
for s in "4 80 256 320 192 1 1 0" "4 32 256 320 192 1 1 0" "4 192 256 320 16 3 1 0" "1 80 256 320 192 1 1 0" "1 192 256 320 16 3 1 0"; do
  for f in "" "2,3,8,16,1" "2,3,8,16,0" "2,3,4,16,1" "2,2,4,16,1" "1,3,4,16,1" "2,3,4,16,0" "2,1,8,16,1" "1,1,4,16,1" "1,1,4,16,0" "1,1,2,16,1" "2,1,4,16,1" "1,1,1,16,1"; do
    echo -n "$s force=[$f] : "; RGBD_CONV_FORCE=$f timeout -k 5 60 python tools/conv_one.py $s 2>/dev/null | tail -1 | awk '{print $4, $NF}'
  done
done
