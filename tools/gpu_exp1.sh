#!/bin/bash
# experiment: the stream threshold kernel -- parity first, then A/B timings
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
( timeout 400 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 ) > $OUT/exp1_pytest.log 2>&1; tail -3 $OUT/exp1_pytest.log
timeout 600 python tools/gpu_ab.py "$@" > $OUT/exp1_ab.log 2>&1
cat $OUT/exp1_ab.log | cut -c1-900
