#!/bin/bash
# bench + kernel stats + PMC traffic for the committed profiles (no tests): outputs under gpurun_out/
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
( time timeout 900 python bench.py ) > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 | cut -c1-400
rm -rf $OUT/prof
( timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ) > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-160
bash tools/gpu_pmc3.sh 2>&1 | tail -30
