"""debug: frames queued ahead in group mode against the counted road"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fiducials_amd import stag as fstag, synth
words = fstag.load_library(21)
w, h = 1280, 720
small = [synth.make_stag_frame(words, 300 + i, w, h, 2).image for i in range(2)]
big = [synth.make_stag_frame(words, 310 + i, w, h, 12).image for i in range(2)]
blank = np.full((h, w), 128, np.uint8)
K = np.array([[933.3, 0, 640.0], [0, 933.3, 360.0], [0, 0, 1]])
frames = np.stack([small[0], big[0], small[1], big[1], blank, big[0], small[0], big[1]])
os.environ["FID_STAG_SPEC"] = "0"
det0 = fstag.StagDetector(21, 7, max_width=w, max_height=h)
want = []
for f in frames:
    m = det0.detect_markers(f)
    want.append(m.copy())
det0.close()
os.environ["FID_STAG_SPEC"] = os.environ.get("SPEC", "1")
pool = fstag.StagPool(21, 7, n_contexts=int(os.environ.get("CTX", "4")), max_width=w, max_height=h)
for rnd in range(4):
    fr = frames if rnd % 2 == 0 else frames[::-1]
    wt = want if rnd % 2 == 0 else want[::-1]
    M, P = pool.detect_markers_batch(fr, K, None, 0.18)
    for f in range(len(fr)):
        ok = M[f].tobytes() == wt[f].tobytes()
        if not ok:
            same_ids = sorted(M[f]["id"].tolist()) == sorted(wt[f]["id"].tolist())
            which = [k for k in range(len(want)) if M[f].tobytes() == want[k].tobytes()]
            print("round", rnd, "frame", f, "MISMATCH: ids got", M[f]["id"].tolist(), "want", wt[f]["id"].tolist(), "same id set", same_ids, "equals want of frame", which)
            if same_ids and len(M[f]) == len(wt[f]):
                o = np.argsort(M[f]["id"]); o2 = np.argsort(wt[f]["id"])
                print("   max corner diff by id", np.abs(M[f]["corners"][o] - wt[f]["corners"][o2]).max())
    print("round", rnd, "stats", [d.queue_stats() for d in pool.dets])
pool.close()
