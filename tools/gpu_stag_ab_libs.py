#!/usr/bin/env python3
"""Two builds of the library on the same random STag frames (three image sizes, 0-20 markers, noise, blank frames): markers, poses
and refusals must be the same bytes frame by frame -- a new kernel against the build whose results were checked against the
reference's code in the rounds before.  Usage: gpu_stag_ab_libs.py <libA.so> <libB.so> [n_frames] [seed]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("AB_CHILD") == "1":
    import numpy as np
    from fiducials_amd import stag as fstag, synth
    from fiducials_amd._lib import FidError
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    words = fstag.load_library(21)
    det = fstag.StagDetector(21, 7, max_width=1920, max_height=1080)
    K = np.array([[1400.0, 0, 960.0], [0, 1400.0, 540.0], [0, 0, 1]])
    out = []
    for i in range(n):
        w, h = ((1920, 1080), (1280, 720), (960, 540))[int(rng.integers(0, 3))]
        kind = rng.random()
        if kind < 0.05:
            img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        elif kind < 0.12:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            img = synth.make_stag_frame(words, 12000 + 7 * seed + i, w, h, int(rng.integers(1, 21))).image
            if rng.random() < 0.3:
                img = np.clip(img.astype(np.int32) + rng.integers(-10, 11, img.shape), 0, 255).astype(np.uint8)
        try:
            m = det.detect_markers(img)
            p = det.pose_last(K, None, 0.18)
            segs = det.edge_segments(validated=True)
            hs = hashlib.sha256(m.tobytes() + p.tobytes() + b"".join(s.tobytes() for s in segs) + det.lines(validated=True).tobytes()).hexdigest()[:16]
            out.append([hs, int(len(m))])
        except FidError as e:
            out.append(["refused %d" % e.status, 0])
    print(json.dumps(out))
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
n = sys.argv[3] if len(sys.argv) > 3 else "60"
seed = sys.argv[4] if len(sys.argv) > 4 else "1"
res = []
for lib in (a, b):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), n, seed], env=dict(os.environ, AB_CHILD="1", FID_LIB=lib), capture_output=True, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("[")]
    if not lines:
        print("no result from", lib, p.stderr[-500:])
        sys.exit(2)
    res.append(json.loads(lines[-1]))
bad = [i for i, (x, y) in enumerate(zip(*res)) if x != y]
print(f"stag A/B of two builds: {n} frames (seed {seed}), markers {sum(x[1] for x in res[0])}, refused {sum(x[0].startswith('refused') for x in res[0])}, mismatches {len(bad)} {bad[:10]}")
sys.exit(1 if bad else 0)
