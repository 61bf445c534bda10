#!/bin/bash
# PMC passes restricted to one kernel (regex $1), small batch to keep box time low
export TMPDIR=/tmp
cd /root/repo
K=${1:-k_walk_full}
OUT=gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
run() {
  timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$K" --pmc $2 --output-format csv -d $OUT/$1 -o p -- python bench.py --batch 64 --unique 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1
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
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run fetch "FETCH_SIZE GRBM_GUI_ACTIVE"
run write "WRITE_SIZE"
