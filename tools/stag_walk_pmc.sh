#!/bin/bash
# SQ counters of the STag single-frame kernels (k_stag_route_walk, k_stag_route_extract, k_stag_refine, ...): what a lone wave per
# workgroup spends its cycles on.  Two passes (own runs, kernel trace only) -> gpurun_out/stag_sq/*.csv summary on stdout
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/stag_sq; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/a -o p -- python tools/dbg_rw.py > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/b -o p -- python tools/dbg_rw.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in 'ab':
    f = glob.glob(f'gpurun_out/stag_sq/{tag}/**/*counter_collection.csv', recursive=True)
    if not f: print(tag, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in ('k_stag_route_walk', 'k_stag_route_extract', 'k_stag_refine', 'k_stag_split_lines', 'k_stag_quads', 'k_stag_validate_lines'):
        if k in acc: print(tag, k, {c: round(v / 2) for c, v in sorted(acc[k].items())})  # (two frames)
PY
