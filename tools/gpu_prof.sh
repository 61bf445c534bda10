#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> gpurun_out/prof2/r_results.db
export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/prof2
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof2 -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof2.log 2>&1
tail -2 gpurun_out/prof2.log | cut -c1-300
