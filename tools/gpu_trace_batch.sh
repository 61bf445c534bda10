#!/bin/bash
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/trb
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/trb -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/trb.log 2>&1
tail -1 gpurun_out/trb.log | cut -c1-200
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/trb/**/*.db',recursive=True)[0])
rows=db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_threshold' in r[0]][-2]
tb=rows[idx][1]
for r in rows[idx:]:
    n=re.sub(r'\(.*','',r[0]).replace('void ','')[:30]
    if 'rocclr' in n: continue
    print(f"{(r[1]-tb)/1e6:8.3f} {(r[2]-tb)/1e6:8.3f} {(r[2]-r[1])/1e6:7.3f} ms  s{r[3]} {n}")
PY
python tools/rocpd_stats.py $(find gpurun_out/trb -name '*.db' | head -1) > gpurun_out/trb_stats.csv
