#!/usr/bin/env python3
"""A/B runs of the cfg 3 batch under different library knobs (environment variables read at fid_create), one process,
frames generated once.  Usage: python tools/gpu_ab.py "FID_THR=tile" "FID_THR_NW=3 FID_THR_SPLIT=0" ...   ('' = defaults)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FID_PROFILE", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from fiducials_amd.detector import ArucoDetector  # noqa: E402
from fiducials_amd.synth import K_DEFAULT  # noqa: E402

if os.environ.get("AB_CHILD") != "1" and len(sys.argv) > 2:
    # one fresh process per configuration: the first context of a process measures ~10 % faster than later ones
    import subprocess

    for cfg in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), cfg], env=dict(os.environ, AB_CHILD="1"))
    sys.exit(0)
B = int(os.environ.get("AB_BATCH", "256"))
STEPS = int(os.environ.get("AB_STEPS", "5"))
frames = bench.make_frames(bench.shard_seeds(0, 1, B))
dev = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
ref = None
for cfg in (sys.argv[1:] or [""]):
    kv = dict(x.split("=", 1) for x in cfg.split()) if cfg.strip() else {}
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        det = ArucoDetector("DICT_5X5_250", max_width=1920, max_height=1080, max_batch=B, max_markers=64, max_candidates=2048, max_contours=int(os.environ.get("FID_BENCH_MAX_CONTOURS", "0")))
        for _ in range(2):
            n = det.detect_markers_device(dev.data_ptr(), B, 1920, 1080, unpack=False)
            det.pose_last(0.14, K_DEFAULT, np.zeros(5), unpack=False)
        torch.cuda.synchronize()
        acc = {}
        t0 = time.perf_counter()
        for _ in range(STEPS):
            n = det.detect_markers_device(dev.data_ptr(), B, 1920, 1080, unpack=False)
            det.pose_last(0.14, K_DEFAULT, np.zeros(5), unpack=False)
            for k, v in det.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / STEPS
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        res = det.detect_markers_device(dev.data_ptr(), B, 1920, 1080)
        sig = [(r[1].tolist(), r[0].tobytes()) for r in res]
        if ref is None:
            ref = sig
        same = sig == ref
        det.close()
        print(json.dumps({"cfg": cfg, "fps": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "markers": int(sum(n)), "same_as_first": same,
                          "stage_ms": {k: round(v, 3) for k, v in acc.items()}}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"cfg": cfg, "error": repr(e)}), flush=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
