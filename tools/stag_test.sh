#!/bin/bash
# STag GPU parity tests only
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
( timeout ${T:-300} python -m pytest tests/test_gpu_stag.py -m gpu -q --timeout 120 "$@" ) > gpurun_out/stag_tests.log 2>&1
tail -25 gpurun_out/stag_tests.log | cut -c1-220
