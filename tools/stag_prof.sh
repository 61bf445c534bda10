#!/bin/bash
# STag front end: parity tests, then kernel times on one 1920x1080 frame (rocprofv3 kernel trace)
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_gpu_stag.py -m gpu -q --timeout 100 ) > gpurun_out/stag_tests.log 2>&1
tail -3 gpurun_out/stag_tests.log | cut -c1-200
rm -rf gpurun_out/prof_stag
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stag -o r -- python -m pytest tests/test_gpu_stag.py -m gpu -q -k "size1" > gpurun_out/prof_stag.log 2>&1
tail -2 gpurun_out/prof_stag.log | cut -c1-200
python tools/rocpd_stats.py gpurun_out/prof_stag > gpurun_out/stag_kernel_stats.csv 2>&1
grep -i stag gpurun_out/stag_kernel_stats.csv | head
