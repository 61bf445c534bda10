#!/bin/bash
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/tr1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tr1 -o t -- python tools/gpu_one_frame.py > gpurun_out/tr1.log 2>&1
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/tr1/**/*.db',recursive=True)[0])
rows=db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
# last call: find last k_threshold
idx=[i for i,r in enumerate(rows) if 'k_threshold' in r[0]][-1]
tb=rows[idx][1]
for r in rows[idx:]:
    n=re.sub(r'\(.*','',r[0]).replace('void ','')[:30]
    print(f"{(r[1]-tb)/1e3:8.1f} {(r[2]-tb)/1e3:8.1f} {(r[2]-r[1])/1e3:7.1f} us  s{r[3]} {n}")
PY
