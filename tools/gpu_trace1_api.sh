#!/bin/bash
# One cfg 2 call with the HIP API calls, the copies and the kernels on one clock: where the host's 0.1 ms beside the kernels goes.
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/tr1a
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d gpurun_out/tr1a -o t -- python tools/gpu_one_frame.py > gpurun_out/tr1a.log 2>&1
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/tr1a/**/*.db',recursive=True)[0])
ev=[]
for n,s,e,st in db.execute("select name,start,end,stream_id from kernels"):
    ev.append((s,e,"K s%s %s"%(st,re.sub(r'\(.*','',n).replace('void ','')[:34])))
for r in db.execute("select name,start,end from memory_copies"):
    ev.append((r[1],r[2],"C "+str(r[0])))
for r in db.execute("select name,start,end from regions"):
    ev.append((r[1],r[2],"  api "+str(r[0])))
ev.sort()
ks=[i for i,x in enumerate(ev) if 'k_threshold' in x[2]]
i0=ks[-1]
# back up to the first api call of this fid_detect: the last hipMemcpy H2D before the threshold
j=i0
while j>0 and ev[i0][0]-ev[j][0] < 150e3: j-=1
tb=ev[i0][0]
for s,e,n in ev[j:]:
    print(f"{(s-tb)/1e3:9.1f} {(e-tb)/1e3:9.1f} {(e-s)/1e3:7.1f}  {n}")
PY
