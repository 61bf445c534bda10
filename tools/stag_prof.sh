#!/bin/bash
# STag pipeline: timing vs the reference on the host, then a rocprofv3 kernel trace of the same loop
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
timeout 200 python tools/stag_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
rm -rf gpurun_out/prof_stag
NO_REF=1 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stag -o r -- python tools/stag_bench.py > gpurun_out/prof_stag.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_stag/r_results.db > gpurun_out/stag_kernel_stats.csv 2>&1
head -14 gpurun_out/stag_kernel_stats.csv | cut -c1-60,150-
