#!/bin/bash
# kernel timeline of ONE context taking 256-frame batches one after the other (the one_at_a_time figure of the bench line)
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/tro
AB_STEPS=3 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tro -o t -- python tools/gpu_ab.py "" > gpurun_out/tro.log 2>&1
grep fps gpurun_out/tro.log | cut -c1-120
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/tro/**/*.db',recursive=True)[0])
rows=db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
thr=[i for i,r in enumerate(rows) if 'k_threshold' in r[0]]
idx=thr[-2]   # the last call: two sub-batches
prev_end=max(r[2] for r in rows[:idx])
tb=rows[idx][1]
print(f"previous call's last kernel ended {(prev_end-tb)/1e6:.3f} ms before this call's first")
for r in rows[idx:]:
    n=re.sub(r'\(.*','',r[0]).replace('void ','')[:30]
    if 'rocclr' in n: continue
    print(f"{(r[1]-tb)/1e6:8.3f} {(r[2]-tb)/1e6:8.3f} {(r[2]-r[1])/1e6:7.3f} ms  s{r[3]} {n}")
PY
