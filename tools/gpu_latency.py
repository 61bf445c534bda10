#!/usr/bin/env python3
"""Single-frame latency (BASELINE cfg 2: one 1920x1080 frame, 20 markers, batch 1) -- what the ROS node's imageCallback
sees: fid_detect on a host frame (PCIe copy included), then fid_pose_last.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.synth import make_frame

d = get_predefined_dictionary("DICT_5X5_250")
K = np.array([[1400, 0, 960], [0, 1400, 540], [0, 0, 1]], float)
D = np.zeros(5)
for mb in (1,):
    det = ArucoDetector(d, device=0, max_width=1920, max_height=1080, max_batch=mb, max_markers=64, max_contours=int(os.environ.get("LAT_MAX_CONTOURS", "0")))
    frames = [make_frame(d, 1000 + i, width=1920, height=1080, n_markers=20).image for i in range(8)]
    for prof in (0, 1):
        ts = []
        for it in range(40):
            img = frames[it % 8]
            t0 = time.perf_counter()
            cor, ids = det.detect_markers(img)
            t1 = time.perf_counter()
            det.pose_last(0.14, K, D)
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
        ts = np.array(ts[8:]) * 1e3
        print(f"host frame -> markers: detect median {np.median(ts[:,0]):.3f} ms (min {ts[:,0].min():.3f}), pose {np.median(ts[:,1]):.3f} ms, n={len(ids)}")
    if os.environ.get("FID_PROFILE"):
        print("   stage ms:", {k: round(v, 3) for k, v in det.stage_ms().items()})
    # device-resident frame
    dev = torch.from_numpy(frames[0]).cuda()
    torch.cuda.synchronize()
    ts = []
    for it in range(40):
        t0 = time.perf_counter()
        det.detect_markers_device(dev.data_ptr(), 1, 1920, 1080)
        t1 = time.perf_counter()
        ts.append(t1 - t0)
    ts = np.array(ts[8:]) * 1e3
    print(f"resident frame -> markers: median {np.median(ts):.3f} ms (min {ts.min():.3f}); launches {det.last_launches()}")

# PCIe-inclusive throughput: the same batch path fed from HOST memory (fid_detect_batch stages through pinned buffers)
B = int(os.environ.get("HOST_BATCH", "64"))
detb = ArucoDetector(d, device=0, max_width=1920, max_height=1080, max_batch=B, max_markers=64)
host = np.stack([frames[i % 8] for i in range(B)])
devb = torch.from_numpy(host).cuda()
torch.cuda.synchronize()
for name, fn in (("host memory", lambda: detb.detect_markers_batch(host)), ("resident   ", lambda: detb.detect_markers_device(devb.data_ptr(), B, 1920, 1080))):
    fn()
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f"batch {B} from {name}: {B / np.median(ts):.0f} frames/s (median of 6)")
