#!/usr/bin/env python3
"""JPEG ingest timing (run on the GPU box): B synthetic 1920x1080 marker frames, encoded as compressed_image_transport does
(libjpeg defaults: 4:2:0, quality 80), decoded on the device to the gray image the detector takes; next to it libjpeg-turbo
itself (Pillow) on one host core.  Usage: python tools/gpu_jpeg_bench.py [batch] [quality]"""
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 80
frames = bench.make_frames(bench.shard_seeds(0, 1, min(B, 32)))
files = []
for k in range(B):
    f = frames[k % len(frames)]
    b = io.BytesIO()
    Image.fromarray(np.stack([f, f, f], -1)).save(b, "JPEG", quality=Q, subsampling=2)
    files.append(b.getvalue())
nbytes = sum(len(f) for f in files)
t = time.perf_counter()
for f in files[:16]:
    np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
cpu = (time.perf_counter() - t) / 16
import torch  # noqa: E402

torch.cuda.init()
from fiducials_amd import jpeg as fj  # noqa: E402

out = {"batch": B, "quality": Q, "file_bytes_mean": nbytes // B, "libjpeg_turbo_ms_per_frame_1core": round(cpu * 1e3, 3)}
for nb in sorted({1, min(B, 16), B}):
    dec = fj.JpegDecoder(max_width=1920, max_height=1080, max_batch=nb)
    for _ in range(2):
        dec.decode(files[:nb], "mono8", to_host=False)
    n = max(3, 64 // nb)
    t = time.perf_counter()
    for _ in range(n):
        dec.decode(files[:nb], "mono8", to_host=False)
    dt = (time.perf_counter() - t) / n
    out[f"batch_{nb}"] = {"ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(nb / dt, 1), "rounds": dec.last_rounds(),
                          "coded_MB_per_s": round(nbytes / B * nb / dt / 1e6, 1)}
    dec.close()
print(json.dumps(out))
