#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only) on a small batch, all kernels
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  FID_SUB_FRAMES=64 timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python bench.py --batch 64 --unique 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r['Kernel_Name'].split('(')[0][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(k, c, "dispatches", len(v), "mean", round(sum(v) / len(v)))
PY
done
