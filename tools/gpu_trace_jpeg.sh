#!/bin/bash
# kernel timeline of the JPEG ingest (one call of B frames): bash tools/gpu_trace_jpeg.sh [B]
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/trj
B=${1:-1}
cat > /tmp/jpeg_one.py <<PY
import sys, io; sys.path.insert(0,'/root/repo')
import numpy as np, bench
from PIL import Image
import torch; torch.cuda.init()
from fiducials_amd import jpeg as fj
B=$B
fr=bench.make_frames(bench.shard_seeds(0,1,min(B,16)))
files=[]
for k in range(B):
    b=io.BytesIO(); f=fr[k%len(fr)]; Image.fromarray(np.stack([f,f,f],-1)).save(b,"JPEG",quality=80,subsampling=2); files.append(b.getvalue())
dec=fj.JpegDecoder(1920,1080,B)
for i in range(4): dec.decode(files,"mono8",to_host=False)
print(dec.last_rounds())
PY
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/trj -o t -- python /tmp/jpeg_one.py > gpurun_out/trj.log 2>&1
tail -1 gpurun_out/trj.log
python - <<'PY'
import sqlite3, glob, re
db=sqlite3.connect(glob.glob('gpurun_out/trj/**/*.db',recursive=True)[0])
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_jpeg_huff<0>' in r[0]][-1]
tb=rows[idx][1]; prev=tb
for r in rows[idx-2:]:
    n=re.sub(r'\(.*','',r[0]).replace('void ','')[:34]
    print(f"{(r[1]-tb)/1e3:8.1f} {(r[2]-r[1])/1e3:8.1f} us gap {(r[1]-prev)/1e3:7.1f}  {n}")
    prev=r[2]
PY
