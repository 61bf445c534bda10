"""Batches in flight: cfg 3 through ONE context (a call returns before the next starts: the latency-bound tail of every batch
has the chip to itself) against TWO / THREE contexts driven by a host thread each (the tail of one batch under the front of the
next).  Prints one JSON line per configuration.  Run on the GPU box: python tools/gpu_inflight.py [steps]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(n_ctx, B, steps, dev_ptr, offset_frac=0.5, shares=None, label=""):
    import torch
    from fiducials_amd.detector import ArucoDetector
    from fiducials_amd.synth import K_DEFAULT

    if shares:
        os.environ["FID_SUB_SHARES"] = shares
    else:
        os.environ.pop("FID_SUB_SHARES", None)
    D = np.zeros(5)
    dets = [ArucoDetector("DICT_5X5_250", device=0, max_width=bench.W, max_height=bench.H, max_batch=B, max_markers=64,
                          max_candidates=2048) for _ in range(n_ctx)]
    found = [0] * n_ctx

    def one(i):
        n = dets[i].detect_markers_device(dev_ptr, B, bench.W, bench.H, unpack=False)
        dets[i].pose_last(bench.FIDUCIAL_LEN, K_DEFAULT, D, unpack=False)
        return sum(n)

    for i in range(n_ctx):
        one(i)
        one(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    one(0)
    step_s = time.perf_counter() - t1

    def worker(i, k):
        if i:
            time.sleep(step_s * offset_frac * i / max(n_ctx - 1, 1) if n_ctx > 1 else 0)
        for _ in range(k):
            found[i] += one(i)

    per = [steps // n_ctx + (1 if i < steps % n_ctx else 0) for i in range(n_ctx)]
    th = [threading.Thread(target=worker, args=(i, per[i])) for i in range(n_ctx)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for d in dets:
        d.close()
    out = {"label": label, "contexts": n_ctx, "batch": B, "steps": steps, "shares": shares, "offset": offset_frac,
           "fps": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "solo_step_ms": round(step_s * 1e3, 3),
           "markers_per_frame": round(sum(found) / (B * steps), 2)}
    print(json.dumps(out), flush=True)
    return out


def main():
    import torch

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    B = 256
    frames = bench.make_frames(bench.shard_seeds(0, 1, B))
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    p = dev.data_ptr()
    sweep = os.environ.get("INFLIGHT_SWEEP", "1:,2:,2:50-50,3:,3:50-50").split(",")
    for item in sweep:
        n, _, sh = item.partition(":")
        run(int(n), B, steps, p, offset_frac=0.5, shares=sh.replace("-", ",") or None, label=f"{n} context(s) in flight")


if __name__ == "__main__":
    main()
