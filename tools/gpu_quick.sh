#!/bin/bash
# quick GPU check: parity tests then a short bench (no CPU baseline)
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python -m pytest tests -m gpu -x -q --timeout 120 ) > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/bench.log
