"""Unit checks of the oracle's building blocks against hand-derived answers (no GPU)."""
import numpy as np

import oracle


def test_box_mean_rounding_fixed_point_equals_exact():
    """boxFilter's ColumnSum<ushort,uchar> fixed-point divide (used when win^2 <= 256: 3, 7, 11, 15) is
    exact rounding for every reachable sum, so the oracle's integer round(sum/area) restates both code paths."""
    SHIFT = 23
    for win in (3, 7, 11, 15):
        d = win * win
        scalef = (1 << SHIFT) / d
        divScale = int(np.floor(scalef))
        frac = scalef - divScale
        divDelta = d // 2
        if frac < 0.5:
            divDelta += 1
        else:
            divScale += 1
        s = np.arange(0, 255 * d + 1, dtype=np.int64)
        fixed = ((s + divDelta) * divScale) >> SHIFT
        exact = (2 * s + d) // (2 * d)
        assert np.array_equal(fixed, exact), win


def test_adaptive_threshold_matches_bruteforce():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    for win in (3, 7, 51):
        r = win // 2
        pad = np.pad(img.astype(np.int64), r, mode="edge")
        ref = np.zeros_like(img)
        for y in range(img.shape[0]):
            for x in range(img.shape[1]):
                s = pad[y:y + win, x:x + win].sum()
                mean = int(np.floor(s / (win * win) + 0.5))
                ref[y, x] = 255 if int(img[y, x]) - mean <= -7 else 0
        assert np.array_equal(oracle.adaptive_threshold(img, win, 7.0), ref)


def _contours(rows):
    m = np.array([[1 if c == "#" else 0 for c in r] for r in rows], dtype=np.uint8) * 255
    cs, holes = oracle.find_contours(m)
    return [c.tolist() for c in cs], holes.tolist()


def test_find_contours_hand_cases():
    # single pixel
    cs, holes = _contours(["...", ".#.", "..."])
    assert cs == [[[1, 1]]] and holes == [0]
    # two diagonal pixels are one 8-connected component: [A, D]
    cs, holes = _contours(["#.", ".#"])
    assert cs == [[[0, 0], [1, 1]]] and holes == [0]
    # '^' shape: the apex is visited twice (pixels on 1-px parts repeat)
    cs, holes = _contours([".#.", "#.#"])
    assert cs == [[[1, 0], [0, 1], [1, 0], [2, 1]]]
    # L shape: outer border counter-clockwise in image coordinates, inner corner skipped on the way back
    cs, holes = _contours(["#.", "##"])
    assert cs == [[[0, 0], [0, 1], [1, 1]]]
    # 3x3 ring: outer border then hole border; findContours returns newest first
    cs, holes = _contours(["###", "#.#", "###"])
    assert holes == [1, 0]
    assert cs[1] == [[0, 0], [0, 1], [0, 2], [1, 2], [2, 2], [2, 1], [2, 0], [1, 0]]
    assert cs[0] == [[0, 1], [1, 0], [2, 1], [1, 2]]  # hole border starts left of the hole, runs clockwise


def test_approx_poly_dp_square_and_line():
    sq = [[x, 0] for x in range(0, 20)] + [[20, y] for y in range(0, 20)] + [[x, 20] for x in range(20, 0, -1)] + \
         [[0, y] for y in range(20, 0, -1)]
    out = oracle.approx_poly_dp(np.array(sq, dtype=np.int32), 0.8)
    assert sorted(map(tuple, out.tolist())) == [(0, 0), (0, 20), (20, 0), (20, 20)]
    line = [[x, 5] for x in range(10)] + [[x, 5] for x in range(9, -1, -1)]
    out = oracle.approx_poly_dp(np.array(line, dtype=np.int32), 1.0)
    assert len(out) == 2


def test_pnp_recovers_synthetic_pose():
    K = np.array([1400.0, 0, 960, 0, 1400.0, 540, 0, 0, 1])
    D = np.array([0.1, -0.2, 0.001, 0.002, 0.0])
    rv = np.array([0.3, -0.2, 1.1])
    tv = np.array([0.1, -0.05, 1.3])
    L = 0.14
    obj = np.array([[-L / 2, L / 2, 0], [L / 2, L / 2, 0], [L / 2, -L / 2, 0], [-L / 2, -L / 2, 0]], dtype=np.float32)
    import ctypes as C
    img = np.zeros(8)
    oracle.lib().ora_project_points(K.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p),
                                    tv.ctypes.data_as(C.c_void_p), obj.ctypes.data_as(C.c_void_p), 4, img.ctypes.data_as(C.c_void_p))
    r, t, e = oracle.solve_pnp_square(K, D, img.astype(np.float32), L)
    assert np.abs(r - rv).max() < 1e-3 and np.abs(t - tv).max() < 1e-3 and e < 1e-6


def test_adaptive_threshold_constant_is_floored_for_binary_inv():
    """thresh.cpp adaptiveThreshold: idelta = type == THRESH_BINARY ? cvCeil(delta) : cvFloor(delta); aruco's _threshold
    passes THRESH_BINARY_INV, whose table is tab[i] = (i - 255 <= -idelta).  A pixel 7 below its box mean is foreground for
    C = 7 and for C = 7.5 (floor -> 7), background for C = 8."""
    img = np.full((20, 20), 100, np.uint8)
    img[10, 10] = 92  # 3 x 3 mean = round(892 / 9) = 99, src - mean = -7
    for c, want in ((7.0, 255), (7.5, 255), (7.999, 255), (8.0, 0), (6.5, 255), (-0.5, 255)):
        out = oracle.adaptive_threshold(img, 3, c)
        assert out[10, 10] == want, (c, out[10, 10])
    # negative constants floor away from zero: C = -0.5 -> idelta = -1: src - mean <= 1 is foreground everywhere on a flat image
    assert oracle.adaptive_threshold(np.full((8, 8), 50, np.uint8), 3, -0.5).min() == 255
    assert oracle.adaptive_threshold(np.full((8, 8), 50, np.uint8), 3, 0.5).min() == 255   # idelta 0: 0 <= 0
    assert oracle.adaptive_threshold(np.full((8, 8), 50, np.uint8), 3, 1.0).max() == 0
