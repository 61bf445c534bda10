"""ctypes wrapper of oracle/_ref/libstag_ref.so: the REFERENCE's own EDPF front-end code (gradient map, anchors, anchor
sort) compiled in place from /root/reference/stag_detect, plus a restatement of the one OpenCV call in front of it
(5x5 Gaussian).  TEST INFRASTRUCTURE ONLY -- imported by tests/, never by fiducials_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.environ.get("ORACLE_SO_DIR") or os.path.join(_HERE, "_ref"), "libstag_ref.so")  # ORACLE_SO_DIR: sanitizer builds
REF_ROOT = "/root/reference/stag_detect"
_LIB = None


def build(force: bool = False) -> str | None:
    """Compile the reference sources where they lie (only where /root/reference is mounted).  Returns the .so path, or None
    when neither the reference tree nor a prebuilt library is available."""
    srcs = [os.path.join(_HERE, "stag_ref.cpp"), os.path.join(_HERE, "cvshim", "opencv2", "opencv.hpp"), os.path.join(_HERE, "Makefile")]
    if os.path.isdir(REF_ROOT) and (force or not os.path.exists(SO) or any(os.path.getmtime(f) > os.path.getmtime(SO) for f in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return SO if os.path.exists(SO) else None


def available() -> bool:
    return build() is not None


def lib():
    global _LIB
    if _LIB is None:
        if build() is None:
            raise RuntimeError("oracle/_ref/libstag_ref.so is missing and /root/reference is not mounted")
        _LIB = C.CDLL(SO)
    return _LIB


def constants():
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().ref_stag_constants(C.byref(a), C.byref(b), C.byref(c))
    return dict(EDGE_VERTICAL=a.value, EDGE_HORIZONTAL=b.value, ANCHOR_PIXEL=c.value)


def smooth5(gray: np.ndarray) -> np.ndarray:
    g = np.ascontiguousarray(gray, dtype=np.uint8)
    out = np.empty_like(g)
    h, w = g.shape
    assert lib().ref_stag_smooth5(g.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w, h) == 0
    return out


def smooth3(gray: np.ndarray) -> np.ndarray:
    g = np.ascontiguousarray(gray, dtype=np.uint8)
    out = np.empty_like(g)
    h, w = g.shape
    assert lib().ref_stag_smooth3(g.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w, h) == 0
    return out


def gradient(smooth: np.ndarray, grad_thresh: int = 16):
    s = np.ascontiguousarray(smooth, dtype=np.uint8)
    h, w = s.shape
    grad = np.zeros((h, w), np.int16)
    dirs = np.zeros((h, w), np.uint8)  # the reference leaves pixels below the threshold unwritten: zero-filled here
    assert lib().ref_stag_gradient(s.ctypes.data_as(C.c_void_p), grad.ctypes.data_as(C.c_void_p),
                                   dirs.ctypes.data_as(C.c_void_p), w, h, grad_thresh) == 0
    return grad, dirs


def anchors(grad: np.ndarray, dirs: np.ndarray, grad_thresh: int = 16, anchor_thresh: int = 0, scan_interval: int = 1):
    g = np.ascontiguousarray(grad, dtype=np.int16)
    d = np.ascontiguousarray(dirs, dtype=np.uint8)
    h, w = g.shape
    edge = np.zeros((h, w), np.uint8)
    cap = w * h
    sorted_ = np.zeros(cap, np.int32)
    n = C.c_int(0)
    rc = lib().ref_stag_anchors(g.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), w, h, grad_thresh, anchor_thresh,
                                scan_interval, edge.ctypes.data_as(C.c_void_p), sorted_.ctypes.data_as(C.c_void_p), cap,
                                C.byref(n))
    assert rc == 0
    return edge, sorted_[:n.value].copy()


def _route_raw(grad, dirs, anchor_map, grad_thresh=16, min_path_len=10):
    g = np.ascontiguousarray(grad, dtype=np.int16)
    d = np.ascontiguousarray(dirs, dtype=np.uint8)
    e = np.ascontiguousarray(anchor_map, dtype=np.uint8).copy()
    h, w = g.shape
    cap = w * h
    pix = np.zeros((cap, 2), np.int32)
    seg = np.zeros((cap // 4 + 16, 2), np.int32)
    ns, npx = C.c_int(0), C.c_int(0)
    rc = lib().ref_stag_route(g.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), w, h, grad_thresh, min_path_len,
                              e.ctypes.data_as(C.c_void_p), pix.ctypes.data_as(C.c_void_p), cap,
                              seg.ctypes.data_as(C.c_void_p), len(seg), C.byref(ns), C.byref(npx))
    assert rc == 0
    return e, pix[:npx.value].copy(), seg[:ns.value].copy()


def validate(smooth2: np.ndarray, segpix: np.ndarray, segs: np.ndarray, div: float = 2.25):
    """The reference's ValidateEdgeSegments on an EdgeMap given as (pixel array, (first, length) pairs).  Returns
    (edge image, validated (first, length) pairs)."""
    s2 = np.ascontiguousarray(smooth2, dtype=np.uint8)
    h, w = s2.shape
    pix = np.ascontiguousarray(segpix, dtype=np.int32).reshape(-1, 2)
    sg = np.ascontiguousarray(segs, dtype=np.int32).reshape(-1, 2)
    edge = np.zeros((h, w), np.uint8)
    out = np.zeros((w * h // 8 + 16, 2), np.int32)
    n = C.c_int(0)
    rc = lib().ref_stag_validate(s2.ctypes.data_as(C.c_void_p), w, h, pix.ctypes.data_as(C.c_void_p), len(pix),
                                 sg.ctypes.data_as(C.c_void_p), len(sg), C.c_double(div), edge.ctypes.data_as(C.c_void_p),
                                 out.ctypes.data_as(C.c_void_p), len(out), C.byref(n))
    assert rc == 0
    return edge, out[:n.value].copy()


def route(grad: np.ndarray, dirs: np.ndarray, anchor_map: np.ndarray, grad_thresh: int = 16, min_path_len: int = 10):
    """The reference's JoinAnchorPointsUsingSortedAnchors.  Returns (edge image, [(n_i, 2) int32 (r, c) per segment])."""
    g = np.ascontiguousarray(grad, dtype=np.int16)
    d = np.ascontiguousarray(dirs, dtype=np.uint8)
    e = np.ascontiguousarray(anchor_map, dtype=np.uint8).copy()
    h, w = g.shape
    cap = w * h
    pix = np.zeros((cap, 2), np.int32)
    seg = np.zeros((cap // 4 + 16, 2), np.int32)
    ns, npx = C.c_int(0), C.c_int(0)
    rc = lib().ref_stag_route(g.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), w, h, grad_thresh, min_path_len,
                              e.ctypes.data_as(C.c_void_p), pix.ctypes.data_as(C.c_void_p), cap,
                              seg.ctypes.data_as(C.c_void_p), len(seg), C.byref(ns), C.byref(npx))
    assert rc == 0
    return e, [pix[a:a + n].copy() for a, n in seg[:ns.value]]


LINE_FIELDS = ("a", "b", "sx", "sy", "ex", "ey", "invert", "segmentNo", "firstPixelIndex", "len")


def fit_lines(src: np.ndarray, segpix: np.ndarray, segs: np.ndarray, validate: bool = False):
    """The reference's SplitSegment2Lines + JoinCollinearLines (+ ValidateLineSegments) on a given EdgeMap.
    Returns (lines float64 [n][10] in LINE_FIELDS order, MIN_LINE_LEN)."""
    im = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = im.shape
    pix = np.ascontiguousarray(segpix, dtype=np.int32).reshape(-1, 2)
    sg = np.ascontiguousarray(segs, dtype=np.int32).reshape(-1, 2)
    cap = (w + h) * 64
    out = np.zeros((cap, 10), np.float64)
    n, mll = C.c_int(0), C.c_int(0)
    rc = lib().ref_stag_fit_lines(im.ctypes.data_as(C.c_void_p), w, h, pix.ctypes.data_as(C.c_void_p), len(pix),
                                  sg.ctypes.data_as(C.c_void_p), len(sg), int(validate), out.ctypes.data_as(C.c_void_p), cap,
                                  C.byref(n), C.byref(mll))
    assert rc == 0
    return out[:n.value].copy(), mll.value


def detect_lines(src: np.ndarray):
    """The reference's DetectLinesByEDPF end to end (only SmoothImage restated).  Returns (lines [n][10], segs, segpix)."""
    im = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = im.shape
    cap = (w + h) * 64
    out = np.zeros((cap, 10), np.float64)
    seg = np.zeros((w * h // 8 + 16, 2), np.int32)
    pix = np.zeros((w * h, 2), np.int32)
    n, ns, npx = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib().ref_stag_detect_lines(im.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p), cap, C.byref(n),
                                     seg.ctypes.data_as(C.c_void_p), len(seg), C.byref(ns),
                                     pix.ctypes.data_as(C.c_void_p), len(pix), C.byref(npx))
    assert rc == 0
    return out[:n.value].copy(), seg[:ns.value].copy(), pix[:npx.value].copy()


def detect_quads(src: np.ndarray):
    """The reference's QuadDetector::detectQuads end to end.  Returns (quads float64 [n][12], number of corner groups)."""
    im = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = im.shape
    out = np.zeros((4096, 12), np.float64)
    n, ng = C.c_int(0), C.c_int(0)
    rc = lib().ref_stag_detect_quads(im.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n),
                                     C.byref(ng))
    assert rc == 0
    return out[:n.value].copy(), ng.value


MARKER_FIELDS = ("id", "corners", "center", "H", "lineInf", "projectiveDistortion")


def detect_markers(src: np.ndarray, library_hd: int = 21, error_correction: int = 7, refine: bool = True):
    """The reference's Stag::detectMarkers end to end (refine=False: stopped in front of PoseRefiner::refineMarkerPose).
    Returns float64 [n][24]: id, corners (8), center (2), H (9), lineInf (3), projectiveDistortion."""
    im = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = im.shape
    out = np.zeros((4096, 24), np.float64)
    n = C.c_int(0)
    rc = lib().ref_stag_detect_markers(im.ctypes.data_as(C.c_void_p), w, h, library_hd, error_correction, int(refine),
                                       out.ctypes.data_as(C.c_void_p), len(out), C.byref(n))
    assert rc == 0
    return out[:n.value].copy()


def host_tables(w: int, h: int):
    """The reference's NFALUT, MIN_LINE_LEN and code sample points for an image size: (lut int32[], min_line_len, locs [72][3])."""
    lut = np.zeros(4096, np.int32)
    n, mll = C.c_int(0), C.c_int(0)
    locs = np.zeros((72, 3))
    assert lib().ref_stag_host_tables(w, h, lut.ctypes.data_as(C.c_void_p), len(lut), C.byref(n), C.byref(mll),
                                      locs.ctypes.data_as(C.c_void_p)) == 0
    return lut[:n.value].copy(), mll.value, locs


def nfa_valid(n: int, k: int, w: int, h: int) -> bool:
    return bool(lib().ref_stag_nfa_valid(n, k, w, h))
