#!/bin/bash
# group-mode kernel stats of the STag batch (64 slots, 128 frames, 3 steps) for the library FID_LIB points at (default: in-tree)
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/sb_${1:-new}; rm -rf $OUT; mkdir -p $OUT
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
STAG_CHILD=1 CTX=64 B=128 STEPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/profs -o r -- python tools/gpu_stag_batch.py > $OUT/profs.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profs -name "*.db" | head -1) > $OUT/stag_kernel_stats.csv
python - <<PY
import csv,re
for r in list(csv.DictReader(open("$OUT/stag_kernel_stats.csv")))[:9]:
    n=re.sub(r'.*<k_stag_(\w+)_fn>.*',r'\1',r['Name'])[:22]
    print(f"{n:22s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {int(r['MinNs'])/1e3:8.1f}  max {int(r['MaxNs'])/1e3:8.1f}")
PY
grep '^{' $OUT/profs.log | tail -1 | cut -c1-80
rm -rf $OUT/profs
