#!/usr/bin/env python3
"""A few single-frame calls (cfg 2) for a kernel trace: rocprofv3 --kernel-trace -- python tools/gpu_one_frame.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.synth import make_frame, K_DEFAULT
d = get_predefined_dictionary("DICT_5X5_250")
det = ArucoDetector(d, max_width=1920, max_height=1080)
frames = [make_frame(d, 1000 + i).image for i in range(2)]
for it in range(6):
    c, ids = det.detect_markers(frames[it % 2])
    det.pose_last(0.14, K_DEFAULT, np.zeros(5))
print(len(ids))
