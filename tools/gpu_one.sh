#!/bin/bash
# run a pytest selection on the GPU box:  bash tools/gpu_one.sh <file> <-k expr>
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
( timeout ${T:-400} python -m pytest "$@" -m gpu -q --timeout 120 ) > gpurun_out/one.log 2>&1
tail -25 gpurun_out/one.log | cut -c1-220
