#!/bin/bash
# kernel-by-kernel durations of one STag group on its own at several group sizes ("$@" = sizes) -> gpurun_out/r6gtrace/
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r6gtrace; rm -rf $OUT; mkdir -p $OUT
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
for G in "$@"; do
  STAG_CHILD=1 NOQ=1 CTX=$G B=$((G*3)) STEPS=1 FID_STAG_GROUP=$G ${EXTRA:-} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$G -o r -- python tools/gpu_stag_batch.py > $OUT/log$G.txt 2>&1
  cp $(find $OUT/p$G -name '*kernel_trace.csv' | head -1) $OUT/trace_g$G.csv; rm -rf $OUT/p$G; grep fps $OUT/log$G.txt | tail -1
done
python - <<'PY'
import csv, collections, re, glob
def nm(r): return re.sub(r'.*<k_stag_(\w+)_fn>.*',r'\1',r['Kernel_Name'])
tabs={}
for p in sorted(glob.glob('gpurun_out/r6gtrace/trace_g*.csv'), key=lambda s:int(re.search(r'_g(\d+)',s).group(1))):
    g=int(re.search(r'_g(\d+)',p).group(1))
    rows=[r for r in csv.DictReader(open(p)) if 'k_stag_batch' in r['Kernel_Name']]
    d=collections.defaultdict(list)
    for r in rows: d[nm(r)].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    cyc=len(d['refine'])
    tabs[g]={k:(sum(v)/cyc/1e3) for k,v in d.items()}
    print('group',g,'cycles',cyc,'kernel ms per cycle',round(sum(tabs[g].values())/1e3,3),'per frame us',round(sum(tabs[g].values())/g,1))
ks=sorted(set(k for t in tabs.values() for k in t), key=lambda k:-max(t.get(k,0) for t in tabs.values()))
print('kernel'.ljust(20),*[f'g{g:>7d}' for g in tabs])
for k in ks[:26]: print(k.ljust(20),*[f'{tabs[g].get(k,0):8.1f}' for g in tabs])
PY
gzip -9 $OUT/*.csv
