"""cv::line(img, pt1, pt2, color, thickness 1, LINE_8) restated in plain Python for the overlay tests (TEST INFRASTRUCTURE ONLY;
OpenCV 4.2 modules/imgproc/src/drawing.cpp: line -> ThickLine -> Line -> LineIterator(connectivity 8, left_to_right) and
clipLine).  The node calls it through aruco::drawDetectedMarkers (aruco_detect.cpp:381-383).  Written from the description of
the iterator (error term, major / minor steps) rather than from the library's C code: an independent second statement."""
from __future__ import annotations

import numpy as np


def cv_round(v) -> int:
    """saturate_cast<int>(float) = cvRound = cvtss2si on x86-64: round half to even; NaN, +-inf and values outside the int range
    give the 'integer indefinite' INT_MIN."""
    v = np.float32(v)
    if not (v >= np.float32(-2147483648.0) and v < np.float32(2147483648.0)):
        return -2147483648
    return int(np.rint(v))


def clip_line(w, h, p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    right, bottom = w - 1, h - 1
    code = lambda x, y: (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8  # noqa: E731
    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))  # (C's conversion truncates toward zero: int())
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line8_pixels(w, h, p1, p2):
    """The pixels cv::line touches, in drawing order."""
    (x1, y1), (x2, y2) = p1, p2
    if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
        ok, (x1, y1), (x2, y2) = clip_line(w, h, (x1, y1), (x2, y2))
        if not ok:
            return []
        if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
            return []  # clipLine's double arithmetic near +-2^31 (the reference writes outside its image there: left out)
    if x2 < x1:  # left to right
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    out = []
    x, y = x1, y1
    if dy > dx:  # y is the major axis: one row per step, a column now and then
        err = dy - 2 * dx
        for _ in range(dy + 1):
            out.append((x, y))
            if err < 0:
                x += 1
                err += 2 * dy
            err -= 2 * dx
            y += sy
    else:
        err = dx - 2 * dy
        for _ in range(dx + 1):
            out.append((x, y))
            if err < 0:
                y += sy
                err += 2 * dx
            err -= 2 * dy
            x += 1
    return out


def draw_detected_markers(bgr: np.ndarray, corners: np.ndarray) -> np.ndarray:
    h, w = bgr.shape[:2]
    for c in np.asarray(corners, dtype=np.float32).reshape(-1, 4, 2):
        for j in range(4):
            p0 = (cv_round(c[j, 0]), cv_round(c[j, 1]))
            p1 = (cv_round(c[(j + 1) % 4, 0]), cv_round(c[(j + 1) % 4, 1]))
            for x, y in line8_pixels(w, h, p0, p1):
                bgr[y, x] = (0, 255, 0)
    return bgr
