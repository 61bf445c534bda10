#!/bin/bash
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/trs2
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/trs2 -o t -- python bench.py --workload stag --streams 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/trs2.log 2>&1
tail -1 gpurun_out/trs2.log | cut -c1-160
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/trs2/**/*.db',recursive=True)[0])
rows=db.execute("select name,count(*),sum(end-start),avg(end-start),max(end-start) from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
t0,t1=db.execute("select min(start),max(end) from kernels").fetchone()
print("span ms",(t1-t0)/1e6,"sum of kernel durations ms",tot/1e6)
for r in rows[:24]:
    print(f"{re.sub(r'\(.*','',r[0])[:40]:40s} n {r[1]:5d} total {r[2]/1e6:8.2f} ms avg {r[3]/1e3:8.1f} us max {r[4]/1e3:8.1f}  {100*r[2]/tot:5.1f}%")
PY
