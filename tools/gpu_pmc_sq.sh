#!/bin/bash
# SQ counter pass (own run, kernel-trace only) on a small batch: what bounds each kernel
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmcsq
rm -rf $OUT; mkdir -p $OUT
FID_SUB_FRAMES=64 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- python bench.py --batch 64 --unique 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
FID_SUB_FRAMES=64 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o p -- python bench.py --batch 64 --unique 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
for d in a b; do
f=$(find $OUT/$d -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'].split('(')[0].replace('void ','')[:30]][r['Counter_Name']].append(float(r['Counter_Value']))
names=sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(30), "n", *[n[-14:].rjust(15) for n in names])
for k in sorted(acc):
    n=len(next(iter(acc[k].values())))
    print(k.ljust(30), n, *[f"{sum(acc[k][c])/max(len(acc[k][c]),1):15.0f}" for c in names])
PY
done
tail -3 $OUT/a.log | cut -c1-300
