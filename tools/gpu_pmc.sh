#!/bin/bash
# PMC passes (own runs, kernel-trace only) for the contour kernels
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
run() { # name, counters
  timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1
  f=$(find $OUT/$1 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()})
PY
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
run sq2 "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
