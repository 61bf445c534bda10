#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only) of the STag pipeline: 12 frames through tools/stag_bench.py
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmc_stag
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  NO_REF=1 timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python tools/stag_bench.py > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee $OUT/$c.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
tot = 0
for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(k, c, "dispatches", len(v), "KiB per frame", round(sum(v) / 12, 1))
print("TOTAL KiB per frame", round(tot / 12, 1))
PY
done
