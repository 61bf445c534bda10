"""How much of the border work of scale s repeats scale s-1?  (round-4 verdict, item 1b; CPU only, uses the oracle.)

For bench frames (seeds 1000 + i) the 13 adaptive-threshold masks and their cv::findContours borders are computed with the
oracle; a border of scale s is IDENTICAL to one of scale s-1 when the point sequences are equal (same start, same order).
Reported per scale and in all:
  * border points on identical borders / all border points  (what a contour-level reuse could skip at best),
  * the same restricted to borders inside the perimeter gate (the ones that are copied, approximated, ...),
  * the same for the pieces between grid-line crossings (seed segments, G = 128): what a segment-level reuse could skip,
  * points on borders whose 32 x 16 mask tiles (1-px dilated) are all unchanged between the two planes (what the per-tile
    dirty map of the verdict's proposal would recognise without walking).
Usage: python tools/cross_scale.py [n_frames] [first_seed]"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from fiducials_amd.dictionary import get_predefined_dictionary  # noqa: E402
from fiducials_amd.synth import make_frame  # noqa: E402

WINS = list(range(3, 52, 4))
G = 128


def pieces(c):
    """Cut a border at the points that lie on a grid line (approximates the seed states of k_seed_walk)."""
    on = np.nonzero(((c[:, 0] % G) == 0) | ((c[:, 1] % G) == 0))[0]
    if len(on) == 0:
        return [c]
    out = []
    for a, b in zip(on, np.r_[on[1:], on[0] + len(c)]):
        idx = np.arange(a, b) % len(c)
        out.append(c[idx])
    return out


def h(a):
    return hashlib.blake2b(np.ascontiguousarray(a, dtype=np.int32).tobytes(), digest_size=12).digest()


def frame_stats(seed, d):
    img = make_frame(d, seed).image
    H, W = img.shape
    lo, hi = 0.1 * max(W, H), 4.0 * max(W, H)
    st = np.zeros((len(WINS), 8), dtype=np.int64)  # pts, pts_same, gate_pts, gate_same, seg_pts_same, tile_clean_pts, n_gate, n_gate_same
    prev_c, prev_s, prev_mask = set(), set(), None
    for s, win in enumerate(WINS):
        mask = oracle.adaptive_threshold(img, win, 7.0)
        cs, _ = oracle.find_contours(mask)
        cur_c, cur_s = set(), set()
        if prev_mask is not None:
            diff = (mask != prev_mask)
            # dilate by one pixel, then reduce to 32 x 16 tiles
            dd = diff.copy()
            dd[1:] |= diff[:-1]; dd[:-1] |= diff[1:]
            d2 = dd.copy()
            d2[:, 1:] |= dd[:, :-1]; d2[:, :-1] |= dd[:, 1:]
            th, tw = (H + 15) // 16, (W + 31) // 32
            pad = np.zeros((th * 16, tw * 32), dtype=bool)
            pad[:H, :W] = d2
            dirty = pad.reshape(th, 16, tw, 32).any(axis=(1, 3))
        for c in cs:
            n = len(c)
            k = h(c)
            cur_c.add(k)
            same = k in prev_c
            gate = lo <= n <= hi
            st[s, 0] += n
            st[s, 1] += n * same
            if gate:
                st[s, 2] += n
                st[s, 3] += n * same
                st[s, 6] += 1
                st[s, 7] += same
            if n >= 32:
                for p in pieces(c):
                    kp = h(p)
                    cur_s.add(kp)
                    if kp in prev_s:
                        st[s, 4] += len(p)
            if prev_mask is not None and gate:
                t = np.unique((c[:, 1] // 16) * 4096 + c[:, 0] // 32)
                if not dirty[t // 4096, t % 4096].any():
                    st[s, 5] += n
        prev_c, prev_s, prev_mask = cur_c, cur_s, mask
    return st


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    d = get_predefined_dictionary("DICT_5X5_250")
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(n, os.cpu_count() or 1)) as ex:
        sts = list(ex.map(frame_stats, range(first, first + n), [d] * n))
    st = np.sum(sts, axis=0)
    tot = st.sum(axis=0)
    rep = {
        "frames": n, "first_seed": first, "grid": G,
        "per_scale": [dict(win=w, border_points=int(r[0]), same_as_prev=round(r[1] / max(r[0], 1), 4),
                           gate_points=int(r[2]), gate_same=round(r[3] / max(r[2], 1), 4),
                           segment_same=round(r[4] / max(r[0], 1), 4), gate_tile_clean=round(r[5] / max(r[2], 1), 4),
                           gate_borders=int(r[6]), gate_borders_same=int(r[7])) for w, r in zip(WINS, st)],
        "all": dict(border_points_per_frame=round(tot[0] / n), same_as_prev=round(tot[1] / tot[0], 4),
                    gate_points_per_frame=round(tot[2] / n), gate_same=round(tot[3] / tot[2], 4),
                    segment_same=round(tot[4] / tot[0], 4), gate_tile_clean=round(tot[5] / tot[2], 4),
                    gate_borders_per_frame=round(tot[6] / n, 1), gate_borders_same=round(tot[7] / tot[6], 4)),
    }
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
