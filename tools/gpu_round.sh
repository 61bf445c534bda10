#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
( time timeout 600 python -m pytest tests -m gpu -x -q --timeout 300 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
( time timeout 600 python bench.py ) > $OUT/bench.log 2>&1
tail -3 $OUT/bench.log
rm -rf $OUT/prof
( timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ) > $OUT/prof.log 2>&1
ls -R $OUT/prof | head -20
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; else python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.csv; fi
head -30 $OUT/kernel_stats.csv | cut -c1-200
# drop the big traces, keep the stats
find $OUT/prof -name '*kernel_trace.csv' -size +8M -delete
nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -5
( timeout 120 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
