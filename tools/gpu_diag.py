#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle comparison (run on the GPU box: gpurun -- python tools/gpu_diag.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FID_PROFILE", "1")
import numpy as np
import oracle
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.synth import make_frame


def compare(det, gray, d, label):
    h, w = gray.shape
    t = time.time()
    cor, ids = det.detect_markers(gray)
    tg = time.time() - t
    t = time.time()
    oids, ocor, tr = oracle.detect(gray, d, trace=True)
    to = time.time() - t
    print(f"== {label}: {w}x{h}  gpu call {tg*1e3:.1f} ms, oracle {to*1e3:.1f} ms")
    print("   counts [starts, contours, cands, filt, acc, mark, ovf]:", det.tap_counts()[0][:7])
    print("   stage ms:", {k: round(v, 3) for k, v in det.stage_ms().items()})
    p = oracle.default_params()
    ns = (p.adaptiveThreshWinSizeMax - p.adaptiveThreshWinSizeMin) // p.adaptiveThreshWinSizeStep + 1
    masks = det.tap_masks(1, ns, h, w)[0]
    ok = True
    for s in range(ns):
        win = p.adaptiveThreshWinSizeMin + s * p.adaptiveThreshWinSizeStep
        om = oracle.adaptive_threshold(gray, win, p.adaptiveThreshConstant) > 0
        nd = int((om != (masks[s] > 0)).sum())
        if nd:
            ok = False
            ys, xs = np.nonzero(om != (masks[s] > 0))
            print(f"   MASK scale {s} win {win}: {nd} differing pixels, first at", xs[0], ys[0])
    print("   masks bit-exact:", ok)
    n_init = len(tr["initial"]["scale"])
    gc = det.tap_candidates(False)[0]
    cnt = det.tap_counts()[0]
    gn = cnt[2]
    same = gn == n_init
    if same:
        for k in ("scale", "contour_size", "is_hole"):
            if not np.array_equal(gc[k][:gn], tr["initial"][k]):
                same = False
                bad = np.nonzero(gc[k][:gn] != tr["initial"][k])[0]
                print(f"   CAND field {k} differs at {bad[:5]}")
        st = np.stack([gc["start_x"][:gn], gc["start_y"][:gn]], 1)
        if not np.array_equal(st, tr["initial"]["start"]):
            same = False
            print("   CAND start differs")
    print(f"   initial candidates: gpu {gn} oracle {n_init} identical(order,size,start): {same}")
    if not same and gn and n_init:
        # set comparison
        gs = set((int(a), int(b), int(c), int(e)) for a, b, c, e in zip(gc['scale'][:gn], gc['start_x'][:gn], gc['start_y'][:gn], gc['contour_size'][:gn]))
        os_ = set((int(a), int(b[0]), int(b[1]), int(e)) for a, b, e in zip(tr['initial']['scale'], tr['initial']['start'], tr['initial']['contour_size']))
        print("   only gpu:", sorted(gs - os_)[:8], " only oracle:", sorted(os_ - gs)[:8])
    gf = det.tap_candidates(True)[0]
    fn = cnt[3]
    n_f = len(tr["filtered"]["scale"])
    samef = fn == n_f and np.array_equal(gf["corners"][:fn].reshape(fn, 4, 2), tr["filtered"]["corners"])
    print(f"   filtered candidates: gpu {fn} oracle {n_f} identical corners: {samef}")
    if samef:
        gb = det.tap_bits()[0][:fn]
        gi = det.tap_ident()[0][:fn]
        print("   bits identical:", np.array_equal(gb, tr["bits"]), " ident identical:", np.array_equal(gi, tr["ident"]))
        if not np.array_equal(gb, tr["bits"]):
            bad = [i for i in range(fn) if not np.array_equal(gb[i], tr["bits"][i])]
            print("   bits differ for", bad[:10])
    pre = det.tap_presubpix()[0][:cnt[5]]
    print("   presubpix ids identical:", np.array_equal(pre["id"], tr["pre_ids"]),
          " corners identical:", pre["corners"].shape[0] == len(tr["pre_ids"]) and np.array_equal(pre["corners"].reshape(-1, 4, 2), tr["pre_corners"]))
    print("   final ids gpu", ids.tolist()[:30], "\n   final ids ora", oids.tolist()[:30])
    if np.array_equal(ids, oids):
        dd = np.abs(cor - ocor).max() if len(ids) else 0.0
        print("   final corners max |diff| =", dd, " exact:", np.array_equal(cor, ocor))
    return cor, ids, ocor, oids


def main():
    d7 = get_predefined_dictionary(7)
    g = np.load(os.path.join(ROOT, "tests/golden/tag_245_246.npz"))["gray"]
    det = ArucoDetector(7, max_width=1920, max_height=1080)
    compare(det, g, d7, "golden tag_245_246")
    g = np.load(os.path.join(ROOT, "tests/golden/bag_4957.npz"))["gray"]
    cor, ids, ocor, oids = compare(det, g, d7, "golden bag_4957")
    import json
    b = json.load(open(os.path.join(ROOT, "tests/golden/golden.json")))["bag_4957"]
    if len(ids):
        pr = det.estimate_pose_single_markers(cor, ids, 0.14, b["K"], b["D"])
        for i in range(len(ids)):
            r, t, e = oracle.solve_pnp_square(b["K"], b["D"], cor[i], 0.14)
            print(f"   pose id {ids[i]}: |dr| {np.abs(pr.rvecs[i]-r).max():.2e} |dt| {np.abs(pr.tvecs[i]-t).max():.2e} derr {abs(pr.image_error[i]-e):.2e}")
    g = np.load(os.path.join(ROOT, "tests/golden/img_403.npz"))["gray"]
    compare(det, g, d7, "golden 403 (clutter)")
    det.close()
    d6 = get_predefined_dictionary(6)
    det = ArucoDetector(6, max_width=1920, max_height=1080)
    for seed in (1000, 1001):
        fr = make_frame(d6, seed)
        compare(det, fr.image, d6, f"synthetic 1080p seed {seed}")
    det.close()
    d0 = get_predefined_dictionary(0)
    det = ArucoDetector(0, max_width=640, max_height=480)
    fr = make_frame(d0, 7, width=640, height=480, n_markers=4, side_range=(60, 110))
    compare(det, fr.image, d0, "synthetic 640x480 4x4_50")


if __name__ == "__main__":
    main()
