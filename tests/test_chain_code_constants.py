"""The chain codes of the contour stage (fid_kernels.hip: dir_dx / dir_dy / code_delta) are OpenCV's border-following directions
(contours.cpp icvCodeDeltas: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE, image y down).  The kernels keep them as two packed 2-bit
tables; this test reads the constants out of the source and checks them against that order, and checks the packed-step arithmetic
the decoders rely on (x + 65536 y is linear: summing packed steps modulo 2^32 walks the packed point) on a random closed walk."""
import os
import re

import numpy as np

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fiducials_amd", "csrc", "fid_kernels.hip")
OPENCV_DELTAS = [(1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1)]


def tables():
    text = open(SRC).read()
    mx = re.search(r"int dir_dx\(int d\) \{ return \(int\)\(\((0x[0-9A-Fa-f]+)u >> \(2 \* d\)\) & 3u\) - 1; \}", text)
    my = re.search(r"int dir_dy\(int d\) \{ return \(int\)\(\((0x[0-9A-Fa-f]+)u >> \(2 \* d\)\) & 3u\) - 1; \}", text)
    assert mx and my, "dir_dx / dir_dy changed their form: update this test with them"
    return int(mx.group(1), 16), int(my.group(1), 16), text


def test_direction_tables_are_opencvs():
    tx, ty, text = tables()
    for c, (dx, dy) in enumerate(OPENCV_DELTAS):
        assert ((tx >> (2 * c)) & 3) - 1 == dx and ((ty >> (2 * c)) & 3) - 1 == dy, c
    # code_delta and codes8_to_points use the same two tables
    assert text.count("0x%04Xu >> c2" % tx) >= 2 and text.count("0x%04Xu >> c2" % ty) >= 2


def test_packed_steps_walk_the_packed_point():
    tx, ty, _ = tables()
    rng = np.random.default_rng(7)
    codes = rng.integers(0, 8, 5000)
    x, y = 4000, 4000  # (inside the 8 191 x 8 191 the library accepts, far enough from 0 for 5 000 steps)
    p = np.uint32(x | (y << 16))
    for c in codes:
        dx, dy = ((tx >> (2 * c)) & 3) - 1, ((ty >> (2 * c)) & 3) - 1
        delta = np.uint32((((tx >> (2 * c)) & 3) | (((ty >> (2 * c)) & 3) << 16)) - 0x10001 & 0xFFFFFFFF)  # code_delta's expression
        p = np.uint32((int(p) + int(delta)) & 0xFFFFFFFF)
        x, y = x + dx, y + dy
        assert 0 <= x < 65536 and 0 <= y < 65536
        assert int(p) == (x | (y << 16))
