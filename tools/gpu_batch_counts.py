#!/usr/bin/env python3
"""Work counters of the contour stage on the first frames of the bench batch (batch context, run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from fiducials_amd.detector import ArucoDetector

B = 64
frames = bench.make_frames(bench.shard_seeds(0, 1, B))
dev = torch.from_numpy(frames).cuda()
det = ArucoDetector("DICT_5X5_250", max_width=1920, max_height=1080, max_batch=B, max_markers=64, max_candidates=2048)
det.detect_markers_device(dev.data_ptr(), B, 1920, 1080, unpack=False)
c = det.tap_counts()[:B].astype(np.float64)
names = {0: "starts", 10: "seeds", 9: "surv1", 7: "survivors", 1: "contours", 11: "contour points", 8: "chunks", 2: "cands", 3: "filtered", 5: "markers"}
print("per frame (mean over %d frames):" % B, ", ".join(f"{n} {c[:, k].mean():.0f}" for k, n in names.items()))
det.close()
