#!/bin/bash
# k_stag_refine phase ticks per marker (library built with -DSR_TIMING into fiducials_amd/lib/dbg/): one cfg 5 frame
cd /root/repo
mkdir -p fiducials_amd/lib/dbg
[ -f fiducials_amd/lib/dbg/libfid_srtiming.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSR_TIMING \
    -o fiducials_amd/lib/dbg/libfid_srtiming.so fiducials_amd/csrc/fid_api.hip 2> /dev/null
FID_LIB=$PWD/fiducials_amd/lib/dbg/libfid_srtiming.so timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, time
from fiducials_amd import stag as fstag, synth
words=fstag.load_library(21)
fr=synth.make_stag_frame(words,100,1920,1080,20).image
det=fstag.StagDetector(21,7,max_width=1920,max_height=1080)
for i in range(2):
    M=det.detect_markers(fr)
print(len(M))
PY
