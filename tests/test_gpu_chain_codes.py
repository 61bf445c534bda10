"""Contour points travel as chain codes (round 6: 4 bits per border step in 32-byte pool chunks, one byte per point in a contour's
dense array, points rebuilt by a prefix sum from the contour's first point -- fid_kernels.hip: code_delta, codes8_to_points,
k_seg_copy).  The frames here are drawn to reach the corners of that layout that ordinary marker frames rarely touch:
  * borders WITHOUT a seed state and longer than two chunks (squares that sit inside one cell of the 128-px seed grid: the
    survivor walk's rows, every chunk through the chunk table);
  * seed segments LONGER than the two chunks a copy record names (ragged edges: > 128 border steps between two grid lines),
    so that a piece starts in the record's chunks and continues in the table's;
  * contours whose length is not a multiple of 8 (the dense array pads every contour to 8 bytes) and a first piece that
    starts in the middle of a segment (the canonical start is rarely a seed state).
Checked as everywhere: the candidate list (scale, contour size, hole flag, first point, corners) `==` the oracle's, through a
single-frame call (32-px grid) and a 16-frame call (128-px grid, the batch forms of the kernels)."""
import numpy as np
import pytest

import oracle
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.dictionary import get_predefined_dictionary

pytestmark = pytest.mark.gpu

W, H = 1280, 720


def shapes_frame(seed):
    rng = np.random.default_rng(seed)
    img = np.full((H, W), 225, np.uint8)
    # squares inside one 128-px grid cell (no pixel on a grid line): seedless borders of 4 * (s - 1) steps
    for j in range(5):
        for k in range(9):
            s = int(rng.integers(40, 100))
            x0 = 128 * k + int(rng.integers(8, 120 - s)) if s < 112 else 128 * k + 8
            y0 = 128 * j + int(rng.integers(8, 120 - s)) if s < 112 else 128 * j + 8
            if y0 + s < H and x0 + s < W and (j + k) % 2 == 0:
                img[y0:y0 + s, x0:x0 + s] = 35
    # big quads with RAGGED edges (every pixel next to an edge flips with probability 1/2: about twice the steps per px, so a
    # seed segment between two grid lines runs to 200 - 300 steps) and a rotated square whose border meets the grid at all angles;
    # each on a cleared patch of its own
    yy, xx = np.mgrid[0:H, 0:W]
    for (cx, cy, half) in ((300, 360, 200), (700, 470, 120)):
        img[(np.abs(xx - cx) <= half + 12) & (np.abs(yy - cy) <= half + 12)] = 225
        inside = (np.abs(xx - cx) <= half) & (np.abs(yy - cy) <= half)
        ring = (np.abs(xx - cx) <= half + 1) & (np.abs(yy - cy) <= half + 1) & ~inside
        img[inside] = 35
        img[ring & (rng.random((H, W)) < 0.5)] = 35
    # ... and COMB edges on the first one (teeth 6 px wide and high, 12 px apart, clear of the grid lines they run along): two
    # border steps per pixel of travel, i.e. ~256 steps between two grid lines -- a seed segment of more than two chunks
    for x in range(100, 500, 12):
        img[154:160, x:x + 6] = 35
    for y in range(160, 560, 12):
        img[y:y + 6, 94:100] = 35
    img[(np.abs(xx - 1040) + np.abs(yy - 300)) <= 212] = 225
    img[(np.abs(xx - 1040) + np.abs(yy - 300)) <= 200] = 35
    return img


def check_candidates(gc, cnt, tr):
    assert cnt[6] == 0, "capacity overflow flags"
    assert cnt[2] == len(tr["initial"]["scale"])
    assert np.array_equal(gc["scale"], tr["initial"]["scale"])
    assert np.array_equal(gc["contour_size"], tr["initial"]["contour_size"])
    assert np.array_equal(gc["is_hole"], tr["initial"]["is_hole"])
    assert np.array_equal(np.stack([gc["start_x"], gc["start_y"]], 1).reshape(-1, 2), tr["initial"]["start"])
    oc = tr["initial"]["corners"].astype(np.float64)
    cross = (oc[:, 1, 0] - oc[:, 0, 0]) * (oc[:, 2, 1] - oc[:, 0, 1]) - (oc[:, 1, 1] - oc[:, 0, 1]) * (oc[:, 2, 0] - oc[:, 0, 0])
    ocr = tr["initial"]["corners"].copy()
    ocr[cross < 0] = ocr[cross < 0][:, [0, 3, 2, 1]]
    assert np.array_equal(gc["corners"].reshape(-1, 4, 2), ocr)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_long_seedless_borders_and_long_segments(seed):
    img = shapes_frame(seed)
    d = get_predefined_dictionary(6)
    _, _, tr = oracle.detect(img, d, trace=True)
    sizes = np.asarray(tr["initial"]["contour_size"])
    assert len(sizes) >= 20 and sizes.max() > 1000 and np.any(sizes % 8 != 0)  # (the frame does what it was drawn for)
    det = ArucoDetector(6, max_width=W, max_height=H, max_batch=1)
    try:
        det.detect_markers(img)
        cnt = det.tap_counts()[0]
        check_candidates(det.tap_candidates(False)[0][:cnt[2]], cnt, tr)
    finally:
        det.close()
    det = ArucoDetector(6, max_width=W, max_height=H, max_batch=16)
    try:
        det.detect_markers_batch(np.stack([img] * 16))
        for f in (0, 7, 15):
            cnt = det.tap_counts()[f]
            check_candidates(det.tap_candidates(False)[f][:cnt[2]], cnt, tr)
    finally:
        det.close()


@pytest.mark.parametrize("mode", ["legacy", "chain"])
def test_the_other_tracing_modes_read_the_same_codes(monkeypatch, mode):
    monkeypatch.setenv("FID_TRACE", mode)
    img = shapes_frame(4)
    d = get_predefined_dictionary(6)
    _, _, tr = oracle.detect(img, d, trace=True)
    det = ArucoDetector(6, max_width=W, max_height=H, max_batch=16)
    try:
        det.detect_markers_batch(np.stack([img] * 16))
        for f in (0, 15):
            cnt = det.tap_counts()[f]
            check_candidates(det.tap_candidates(False)[f][:cnt[2]], cnt, tr)
    finally:
        det.close()
