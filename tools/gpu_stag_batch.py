#!/usr/bin/env python3
"""STag throughput of fid_stag_detect_markers_batch on the cfg 5 bench frames (run on the GPU box): groups (frames as a grid
dimension) against round 2's stream-per-context road, results compared frame by frame.  Usage: gpu_stag_batch.py [configs...]
where a config is "CTX=32 GROUP=16" style environment settings (FID_STAG_GROUP, FID_STAG_BATCH)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("STAG_CHILD") == "1":
    import numpy as np
    import bench
    if os.environ.get("NOQ") == "1":  # the runtime's default number of hardware queues (bench.py asks for 24 for the aruco path)
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
    from fiducials_amd import stag as fstag, synth
    nctx = int(os.environ.get("CTX", "32"))
    frames = bench.make_stag_frames(bench.shard_seeds(0, 1, bench.STAG_UNIQUE, "stag"))
    B = int(os.environ.get("B", "64"))
    batch = np.stack([frames[i % len(frames)] for i in range(B)])
    pool = fstag.StagPool(bench.STAG_HD, bench.STAG_EC, n_contexts=nctx)
    m, p = pool.detect_markers_batch(batch, synth.K_DEFAULT, None, 0.18)
    t = time.perf_counter()
    steps = int(os.environ.get("STEPS", "4"))
    for _ in range(steps):
        m, p = pool.detect_markers_batch(batch, synth.K_DEFAULT, None, 0.18)
    dt = time.perf_counter() - t
    sig = [(a["id"].tolist(), a["corners"].tobytes(), b["tvec"].tobytes()) for a, b in zip(m, p)]
    import hashlib
    h = hashlib.sha256(repr(sig).encode()).hexdigest()[:16]
    print(json.dumps({"fps": round(B * steps / dt, 1), "markers_per_frame": round(sum(len(a) for a in m) / B, 2), "sig": h}))
    pool.close()
else:
    for cfg in sys.argv[1:] or [""]:
        env = dict(os.environ, STAG_CHILD="1", **dict(kv.split("=", 1) for kv in cfg.split()))
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(cfg or "(default)", "|", line[-1] if line else p.stderr[-600:], "|", "; ".join(l for l in p.stderr.splitlines() if l.startswith("fid stag batch"))[-330:])
