#!/usr/bin/env python3
"""STag (BASELINE cfg 5): one 1920x1080 frame with HD21 markers through fid_stag_detect_markers + fid_stag_pose_last, timed
next to the reference's own Stag::detectMarkers (oracle/_ref, CPU, one thread).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fiducials_amd import stag as fstag, synth
from oracle import stag_ref

hd, ec = 21, 7
words = fstag.load_library(hd)
frames = [synth.make_stag_frame(words, 100 + i, 1920, 1080, 20).image for i in range(4)]
K = np.array([[1400.0, 0, 960.0], [0, 1400.0, 540.0], [0, 0, 1]])
det = fstag.StagDetector(hd, ec, max_width=1920, max_height=1080)
ts = []
for it in range(12):
    img = frames[it % 4]
    t0 = time.perf_counter()
    M = det.detect_markers(img)
    t1 = time.perf_counter()
    P = det.pose_last(K, None, 0.18)
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
ts = np.array(ts[4:]) * 1e3
print(f"GPU  : detect_markers median {np.median(ts[:,0]):.2f} ms (min {ts[:,0].min():.2f}), pose {np.median(ts[:,1]):.3f} ms, markers {len(M)}")
if stag_ref.available() and not os.environ.get("NO_REF"):
    tr = []
    for it in range(6):
        t0 = time.perf_counter()
        R = stag_ref.detect_markers(frames[it % 4], hd, ec, refine=True)
        tr.append(time.perf_counter() - t0)
    tr = np.array(tr[2:]) * 1e3
    print(f"CPU  : reference Stag::detectMarkers (1 thread, incl. Stag construction) median {np.median(tr):.1f} ms, markers {len(R)}")
