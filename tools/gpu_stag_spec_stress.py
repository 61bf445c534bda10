#!/usr/bin/env python3
"""Randomised check of STag frames queued ahead of their own counts (FID_STAG_SPEC, fid_stag.hip) against the counted road: a
sequence of unrelated frames -- marker counts 0..12, two image sizes in turn, noise frames, blank frames -- through one context on
each road; markers and poses must be the same bytes frame by frame, and a frame one road refuses (a noise frame whose walk
passes 32 767 chains: FID_E_CAPACITY) the other road refuses with the same status.
Usage: gpu_stag_spec_stress.py [n_frames] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fiducials_amd import stag as fstag, synth
from fiducials_amd._lib import FidError

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
words = fstag.load_library(21)
frames = []
for i in range(n):
    w, h = ((1280, 720), (960, 540))[int(rng.random() < 0.3)]
    kind = rng.random()
    if kind < 0.08:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    elif kind < 0.16:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = synth.make_stag_frame(words, 9000 + i, w, h, int(rng.integers(1, 13))).image
        if rng.random() < 0.3:
            img = np.clip(img.astype(np.int32) + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)
    frames.append(img)
K = np.array([[933.3, 0, 640.0], [0, 933.3, 360.0], [0, 0, 1]])
res = {}
for road in ("0", "1"):
    os.environ["FID_STAG_SPEC"] = road
    det = fstag.StagDetector(21, 7, max_width=1280, max_height=720)
    out = []
    for f in frames:
        try:
            m = det.detect_markers(f)
            out.append((m.tobytes(), det.pose_last(K, None, 0.18).tobytes(), len(m)))
        except FidError as e:
            out.append(("refused", e.status, 0))
    res[road] = (out, det.queue_stats())
    det.close()
bad = [i for i in range(n) if res["0"][0][i] != res["1"][0][i]]
refused = [i for i in range(n) if res["0"][0][i][0] == "refused"]
print(f"stag queued-ahead stress: {n} frames, markers found {sum(x[2] for x in res['0'][0])}, refused {len(refused)} "
      f"(status {sorted(set(res['0'][0][i][1] for i in refused))}), (queued, rerun) = {res['1'][1]}, mismatches {len(bad)} {bad[:10]}")
sys.exit(1 if bad else 0)
