#!/bin/bash
# SQ counters of the STag batch (one group of 32 frame slots, counter runs serialise the dispatches: a kernel ALONE on the chip):
# VALU wave-instructions and lane utilisation per kernel and frame -> gpurun_out/stag_sq/summary.txt
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/stag_sq; rm -rf $OUT; mkdir -p $OUT
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
STAG_CHILD=1 NOQ=1 CTX=32 B=64 STEPS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/c -o p -- python tools/gpu_stag_batch.py > $OUT/c.log 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/stag_sq/c/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    k = re.sub(r'k_stag_batch<(\w+)_fn>', r'\1[g]', k)
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); calls[k] += 1
frames = 64 * 2  # (the warm-up call and the timed one)
rows = []
for k, v in acc.items():
    valu = v.get('SQ_INSTS_VALU', 0) / frames
    lu = v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']) if v.get('SQ_ACTIVE_INST_VALU') else 0  # (as tools/sq_summary.py)
    rows.append((valu, k, lu, v.get('SQ_INSTS_SALU', 0) / frames, v.get('SQ_WAVE_CYCLES', 0) / frames, calls[k]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open('gpurun_out/stag_sq/summary.txt', 'w') as o:
    for line in [f"STag batch (CTX=32: groups of 16 frame slots), per frame: VALU wave-instructions {tot/1e6:.2f} M"] + [
            f"{k[:38]:38s} VALU {valu/1e6:7.3f} M  lanes {lu:5.2f}  SALU {salu/1e6:6.3f} M  wave-cycles {wc/1e6:7.2f} M  dispatches {c}" for valu, k, lu, salu, wc, c in rows[:30]]:
        print(line); o.write(line + "\n")
PY
rm -rf $OUT/c
