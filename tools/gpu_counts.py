#!/usr/bin/env python3
"""Per-frame work counters of the contour stage on bench frames (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FID_PROFILE", "1")
import numpy as np
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.synth import make_frame

d = get_predefined_dictionary("DICT_5X5_250")
det = ArucoDetector(d, device=0, max_width=1920, max_height=1080, max_batch=1, max_markers=64)
for seed in (1000, 1001, 1002):
    fr = make_frame(d, seed, width=1920, height=1080, n_markers=20)
    cor, ids = det.detect_markers(fr.image)
    c = det.tap_counts()[0]
    print(f"seed {seed}: starts {c[0]} seeds {c[10]} surv1 {c[9]} survivors {c[7]} slots {c[1]} chunks {c[8]} (= {c[8]*64} pts max) cands {c[2]} filt {c[3]} markers {c[5]} ovf {c[6]}")
    print("   stage ms:", {k: round(v, 3) for k, v in det.stage_ms().items()})
