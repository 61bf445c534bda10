#!/bin/bash
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/trs
cat > /tmp/stag_one.py <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, time
from fiducials_amd import stag as fstag, synth
words=fstag.load_library(21)
fr=synth.make_stag_frame(words,100,1920,1080,20).image
det=fstag.StagDetector(21,7,max_width=1920,max_height=1080)
for i in range(6):
    t=time.perf_counter(); M=det.detect_markers(fr); P=det.pose_last(synth.K_DEFAULT,None,0.18); print(len(M), (time.perf_counter()-t)*1e3)
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/trs -o t -- python /tmp/stag_one.py > gpurun_out/trs.log 2>&1
tail -3 gpurun_out/trs.log
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/trs/**/*.db',recursive=True)[0])
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_stag_smooth_grad' in r[0]][-1]
tb=rows[idx][1]; prev=tb; busy=0
for r in rows[idx:]:
    n=re.sub(r'\(.*','',r[0]).replace('void ','')[:34]
    print(f"{(r[1]-tb)/1e3:8.1f} {(r[2]-r[1])/1e3:7.1f} us gap {(r[1]-prev)/1e3:6.1f}  {n}")
    prev=r[2]; busy+=r[2]-r[1]
print("span us", (prev-tb)/1e3, "busy us", busy/1e3)
PY
