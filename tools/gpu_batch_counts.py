#!/usr/bin/env python3
"""Per-frame work spread of the contour stage over the bench batch (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FID_PROFILE", "1")
import numpy as np, torch
import bench
from fiducials_amd.detector import ArucoDetector
B = 256
fr = bench.make_frames([1000 + i for i in range(B)])
d = torch.from_numpy(fr).cuda()
det = ArucoDetector("DICT_5X5_250", device=0, max_batch=B, max_markers=64)
det.detect_markers_device(d.data_ptr(), B, 1920, 1080, unpack=False)
c = det.tap_counts()
for name, col in (("starts", 0), ("surv1", 9), ("surv", 7), ("chunks", 8), ("cands", 2)):
    v = c[:, col]
    print(f"{name}: min {v.min()} mean {v.mean():.0f} max {v.max()}  max/mean {v.max()/v.mean():.2f}")
print({k: round(v, 3) for k, v in det.stage_ms().items()})
