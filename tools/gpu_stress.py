#!/usr/bin/env python3
"""Randomised parity sweep of the aruco path against the oracle (run on the GPU box): frames of odd and even sizes, small and
large markers (borders that cross many / no seed grid lines), noise levels, rectangle clutter, through single-frame calls
(32 px seed grid, 256 walker workgroups), batch calls (64 / 128 px grids) and calls of >= 16 frames (the batch forms of the kernels);
ids and corners must be `==` the oracle's.
Usage: python tools/gpu_stress.py [n_cases] [first_seed]"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from fiducials_amd.dictionary import get_predefined_dictionary  # noqa: E402
from fiducials_amd.synth import make_frame  # noqa: E402

SIZES = [(1920, 1080), (1280, 720), (1000, 700), (641, 479), (1283, 517), (800, 600), (333, 555)]


def gen(seed):
    rng = np.random.default_rng(seed)
    W, H = SIZES[seed % len(SIZES)]
    d = get_predefined_dictionary(6)
    big = rng.random() < 0.4
    nm = int(rng.integers(1, 5)) if big else int(rng.integers(4, 24))
    lo = float(rng.uniform(150, 300)) if big else float(rng.uniform(40, 110))
    scale = min(W / 1920.0, H / 1080.0) ** 0.5
    noise, tilt = float(rng.uniform(0, 5)), float(rng.uniform(5, 50))
    try:
        fr = make_frame(d, 50000 + seed, width=W, height=H, n_markers=nm, noise_sigma=noise,
                        side_range=(lo * scale, lo * scale * 1.6), max_tilt_deg=tilt)
    except ValueError:  # (markers too large for the layout grid of this size)
        fr = make_frame(d, 50000 + seed, width=W, height=H, n_markers=2, noise_sigma=noise, side_range=(60.0, 90.0), max_tilt_deg=tilt)
    img = fr.image.copy()
    if rng.random() < 0.5:  # rectangle clutter: long borders without markers, nested boxes
        for _ in range(int(rng.integers(5, 60))):
            x, y = int(rng.integers(0, W - 8)), int(rng.integers(0, H - 8))
            w, h = int(rng.integers(2, W // 3)), int(rng.integers(2, H // 3))
            img[y:y + h, x:x + w] = rng.choice([20, 60, 200, 240])
    return img


def ora(img):
    d = get_predefined_dictionary(6)
    ids, corners = oracle.detect(img, d)
    return ids.tolist(), np.asarray(corners)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 84
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    with mp.get_context("fork").Pool(min(n, os.cpu_count() or 1, 96)) as pool:
        imgs = pool.map(gen, range(s0, s0 + n), chunksize=1)
        oras = pool.map(ora, imgs, chunksize=1)
    from fiducials_amd.detector import ArucoDetector  # (after the fork pool: no HIP context in the children)

    bad = 0
    by_size = {}
    for i, im in enumerate(imgs):
        by_size.setdefault(im.shape, []).append(i)
    from fiducials_amd._lib import FidError

    BIG = dict(max_starts=1 << 20, max_contours=131072, max_points=32 << 20, max_candidates=4096)
    retries = 0

    def run(make, call):
        """call(det) on a default-limits context; a reported capacity overflow (never a silent one) is retried with big tables"""
        nonlocal retries
        det = make({})
        try:
            return call(det)
        except FidError as e:
            if e.status != 4:
                raise
            retries += 1
            print("  capacity (retrying with larger tables):", e, flush=True)
            det.close()
            det = make(BIG)
            return call(det)
        finally:
            det.close()

    for (H, W), idx in by_size.items():
        res_b = run(lambda kw: ArucoDetector(6, max_width=W, max_height=H, max_batch=len(idx), **kw),
                    lambda det: det.detect_markers_batch(np.stack([imgs[i] for i in idx])))
        # the same frames again as a call of >= 16 frames: from there on a call runs the BATCH forms of the kernels (one survivor-walk
        # workgroup a frame, k_probe_refill, the queued k_near) -- they must see random content too
        rep = -(-16 // len(idx))
        res_r = run(lambda kw: ArucoDetector(6, max_width=W, max_height=H, max_batch=rep * len(idx), **kw),
                    lambda det: det.detect_markers_batch(np.stack([imgs[i] for i in idx] * rep)))
        for k, i in enumerate(idx):
            oi, oc = oras[i]
            for q in range(rep):
                r = res_r[q * len(idx) + k]
                if not (r[1].tolist() == oi and np.array_equal(np.asarray(r[0]).reshape(oc.shape), oc)):
                    bad += 1
                    print(f"MISMATCH case {s0 + i} size {W}x{H}: copy {q} of the {rep * len(idx)}-frame call, ids {r[1].tolist()} vs {oi}", flush=True)
        for k, i in enumerate(idx):
            c1, id1 = run(lambda kw: ArucoDetector(6, max_width=W, max_height=H, max_batch=1, **kw), lambda det: det.detect_markers(imgs[i]))
            oi, oc = oras[i]
            ok1 = id1.tolist() == oi and np.array_equal(np.asarray(c1).reshape(oc.shape), oc)
            ok2 = res_b[k][1].tolist() == oi and np.array_equal(np.asarray(res_b[k][0]).reshape(oc.shape), oc)
            if not (ok1 and ok2):
                bad += 1
                print(f"MISMATCH case {s0 + i} size {W}x{H}: single {ok1} batch {ok2} ids {id1.tolist()} vs {oi}", flush=True)
        print(f"{W}x{H}: {len(idx)} frames, markers found {sum(len(oras[i][0]) for i in idx)}", flush=True)
    print("capacity overflows reported and retried with larger tables:", retries)
    print("stress:", n, "cases,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
