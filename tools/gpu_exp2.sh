#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for B in 64 128 256; do
  AB_BATCH=$B timeout 300 python tools/gpu_ab.py "FID_SUB_FRAMES=$B" 2>&1 | grep cfg | cut -c1-700
done
