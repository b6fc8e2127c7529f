#!/bin/bash
# a few representative conv shapes of the c2 workload, kernel-only timing (n cin h w cout k stride transposed)
for s in "8 96 128 128 96 3 1 0" "8 192 128 128 96 1 1 0" "8 96 128 128 192 1 1 0" "8 96 64 64 96 3 1 0" "8 192 64 64 96 1 1 0" \
         "8 384 128 128 192 5 2 0" "8 192 64 64 192 5 2 0" "8 320 16 16 192 5 2 1" "8 192 32 32 192 5 2 1" "8 192 64 64 192 5 2 1" \
         "8 16 256 256 192 5 2 0" "8 384 128 128 192 3 1 0" "8 384 128 128 192 1 1 0" "8 224 16 16 128 5 1 0" "8 1280 16 16 213 1 1 0"; do
  echo -n "$s : "; timeout -k 5 60 python tools/conv_one.py $s 2>/dev/null | tail -1
done
