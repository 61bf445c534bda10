#!/bin/bash
# full GPU test suite + randomised stress sweep + default bench line
set -u
export TMPDIR=/tmp
cd /root/repo; OUT=gpurun_out/r3g; mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
( timeout 600 python tools/gpu_stress.py ${STRESS_N:-150} ) > $OUT/stress.log 2>&1; tail -4 $OUT/stress.log
( time timeout 900 python bench.py ) > $OUT/bench.log 2>&1
tail -4 $OUT/bench.log | cut -c1-3000
