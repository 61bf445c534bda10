#!/usr/bin/env python3
"""Where the JPEG stream -> markers time goes: the decoder thread's per-batch wall time alone and beside the detector,
the detector's alone, and the main thread's waits.  Run on the GPU box."""
import io, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
import bench
from fiducials_amd import jpeg as fj
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.synth import K_DEFAULT

B = int(os.environ.get("JB", "256"))
frames = bench.make_frames(bench.shard_seeds(0, 1, min(B, 64)))
files = []
for k in range(B):
    b = io.BytesIO()
    Image.fromarray(np.stack([frames[k % len(frames)]] * 3, -1)).save(b, "JPEG", quality=80, subsampling=2)
    files.append(b.getvalue())
torch.cuda.init()
J = int(os.environ.get('JJ', '3'))
NT = int(os.environ.get('JT', '1'))
decs = [fj.JpegDecoder(max_width=1920, max_height=1080, max_batch=B, device=0) for _ in range(J)]
dets = [ArucoDetector("DICT_5X5_250", device=0, max_width=1920, max_height=1080, max_batch=B, max_markers=64, max_candidates=2048) for _ in range(2)]
D = np.zeros(5)
for d in decs:
    d.decode(files, "mono8", to_host=False)
# decoder alone
t = time.perf_counter()
for k in range(6):
    decs[k % J].decode(files, "mono8", to_host=False)
print(f"decoder alone: {(time.perf_counter() - t) / 6 * 1e3:.2f} ms per batch")
# detector alone, two contexts in turn, on decoded frames
ptr, w, h, _, _ = decs[0].device_ptr()
def det_alone(n):
    for k in range(n):
        if k >= 2:
            dets[k % 2].collect(unpack=False); dets[k % 2].pose_last(0.14, K_DEFAULT, D, unpack=False)
        dets[k % 2].submit_device(ptr, B, w, h, after=dets[(k - 1) % 2])
    for k in range(max(n - 2, 0), n):
        dets[k % 2].collect(unpack=False); dets[k % 2].pose_last(0.14, K_DEFAULT, D, unpack=False)
det_alone(3)
t = time.perf_counter(); det_alone(6)
print(f"detector alone: {(time.perf_counter() - t) / 6 * 1e3:.2f} ms per batch")
# both
def run(n):
    ready = [threading.Event() for _ in range(n)]; freed = [threading.Event() for _ in range(n)]
    dec_ms, wait_ms, sub_ms, col_ms = [], [], [], []
    def decode_side(t):
        for k in range(t, n, NT):
            if k >= J: freed[k - J].wait()
            t0 = time.perf_counter(); decs[k % J].decode(files, "mono8", to_host=False); dec_ms.append((time.perf_counter() - t0) * 1e3)
            ready[k].set()
    ths = [threading.Thread(target=decode_side, args=(t,)) for t in range(NT)]
    for th in ths: th.start()
    def collect(k):
        t0 = time.perf_counter()
        dets[k % 2].collect(unpack=False); dets[k % 2].pose_last(0.14, K_DEFAULT, D, unpack=False); freed[k].set()
        col_ms.append((time.perf_counter() - t0) * 1e3)
    for k in range(n):
        t0 = time.perf_counter(); ready[k].wait(); wait_ms.append((time.perf_counter() - t0) * 1e3)
        if k >= 2: collect(k - 2)
        p, w_, h_, _, _ = decs[k % J].device_ptr()
        t0 = time.perf_counter(); dets[k % 2].submit_device(p, B, w_, h_, after=dets[(k - 1) % 2]); sub_ms.append((time.perf_counter() - t0) * 1e3)
    for k in range(max(n - 2, 0), n): collect(k)
    for th in ths: th.join()
    return dec_ms, wait_ms, sub_ms, col_ms
run(3)
t = time.perf_counter(); r = run(12); dt = time.perf_counter() - t
print(f"both (J={J}, {NT} decoder threads): {dt / 12 * 1e3:.2f} ms per batch = {B * 12 / dt:.0f} frames/s")
for name, v in zip(("decode call", "wait for decoded", "submit", "collect"), r):
    print(f"  {name:18s} ms:", " ".join(f"{x:.1f}" for x in v))
