#!/bin/bash
# kernel-trace statistics of the bench's two-context run for one library build: gpu_stats_lib.sh lib.so kernel-name-pattern
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/stl
FID_LIB=$1 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/stl -o r -- python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/stl.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/stl -name '*.db' | head -1) | grep -E "$2" | cut -c1-40,120-
tail -1 gpurun_out/stl.log | cut -c1-150
