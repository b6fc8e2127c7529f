#!/bin/bash
# Knock-out timings of the conv main loop.  The diagnostic builds are not kept in the tree: wrap the
# statements in conv_mfma.hip in `#if !(RGBD_DIAG & bit)` -- bit 0: dma_w() of the next stage, bit 1: dma_p() of the next
# chunk, bit 2: the per-stage __syncthreads(), bit 3: the af/bf fragment reads after the first tap -- compile
# conv_mfma.hip with -DRGBD_DIAG=n, link ab/diag<n>.so with the other objects of csrc/build/, and run this script on the
# GPU box.  Results of such builds are garbage; only the kernel-only timing means something.
for shape in "8 96 128 128 96 3 1 0" "4 192 256 320 192 3 1 0" "4 192 256 320 192 5 2 0"; do
  for d in ${DIAGS:-0 1 2 3 7 11 15}; do
    echo -n "shape [$shape] diag$d: "
    RGBD_AMD_LIB=$PWD/ab/diag$d.so timeout -k 5 120 python tools/conv_one.py $shape 2>&1 | tail -1
  done
done
