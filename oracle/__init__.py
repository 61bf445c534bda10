"""ctypes wrapper of the parity oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by fiducials_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OraParams(C.Structure):
    _fields_ = [
        ("adaptiveThreshConstant", C.c_double),
        ("adaptiveThreshWinSizeMin", C.c_int),
        ("adaptiveThreshWinSizeMax", C.c_int),
        ("adaptiveThreshWinSizeStep", C.c_int),
        ("cornerRefinementMethod", C.c_int),
        ("cornerRefinementWinSize", C.c_int),
        ("cornerRefinementMaxIterations", C.c_int),
        ("cornerRefinementMinAccuracy", C.c_double),
        ("errorCorrectionRate", C.c_double),
        ("minCornerDistanceRate", C.c_double),
        ("markerBorderBits", C.c_int),
        ("maxErroneousBitsInBorderRate", C.c_double),
        ("minDistanceToBorder", C.c_int),
        ("minMarkerDistanceRate", C.c_double),
        ("minMarkerPerimeterRate", C.c_double),
        ("maxMarkerPerimeterRate", C.c_double),
        ("minOtsuStdDev", C.c_double),
        ("perspectiveRemoveIgnoredMarginPerCell", C.c_double),
        ("perspectiveRemovePixelPerCell", C.c_int),
        ("polygonalApproxAccuracyRate", C.c_double),
    ]


class OraDict(C.Structure):
    _fields_ = [("marker_size", C.c_int), ("max_correction_bits", C.c_int), ("n_markers", C.c_int),
                ("bytes", C.c_void_p)]


class OraMarker(C.Structure):
    _fields_ = [("id", C.c_int32), ("corners", C.c_float * 8)]


class OraCandidate(C.Structure):
    _fields_ = [("scale", C.c_int32), ("contour_size", C.c_int32), ("start_x", C.c_int32),
                ("start_y", C.c_int32), ("is_hole", C.c_int32), ("corners", C.c_float * 8)]


class OraTrace(C.Structure):
    _fields_ = [
        ("initial", C.POINTER(OraCandidate)), ("cap_initial", C.c_int), ("n_initial", C.c_int),
        ("filtered", C.POINTER(OraCandidate)), ("cap_filtered", C.c_int), ("n_filtered", C.c_int),
        ("bits", C.POINTER(C.c_uint8)),
        ("ident", C.POINTER(C.c_int32)),
        ("presubpix", C.POINTER(OraMarker)), ("cap_pre", C.c_int), ("n_pre", C.c_int),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("aruco_detect_oracle.c", "pnp_oracle.c", "aruco_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def build_native(out_dir: str) -> str:
    """The same two C files built for the machine they run on (-O3 -march=native; -ffp-contract=off keeps the IEEE operation
    sequence, so results do not change).  For bench.py's cpu_baseline leg on the GPU box's host (SURVEY.md 8d asks for
    -march=native there); the shipped liboracle.so stays generic because it travels between machines."""
    so = os.path.join(out_dir, "liboracle_native.so")
    srcs = [os.path.join(_HERE, f) for f in ("aruco_detect_oracle.c", "pnp_oracle.c")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off",
                               "-fno-fast-math", "-shared", "-o", so] + srcs + ["-lm"])
    return so


def use_library(path: str | None):
    """Point this wrapper at another build of the same sources (build_native); None = the shipped liboracle.so."""
    global _LIB, _SO_OVERRIDE
    _SO_OVERRIDE = path
    _LIB = None


_SO_OVERRIDE = None


def lib():
    global _LIB
    if _LIB is None:
        so = _SO_OVERRIDE or os.path.join(os.environ.get("ORACLE_SO_DIR") or _HERE, "liboracle.so")  # ORACLE_SO_DIR: sanitizer builds
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.ora_fiducial_area.restype = C.c_double
    return _LIB


def default_params() -> OraParams:
    p = OraParams()
    lib().ora_default_params(C.byref(p))
    return p


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def make_dict(d) -> OraDict:
    """d: fiducials_amd.dictionary.Dictionary.  Keeps a reference to the byte table on the struct."""
    od = OraDict(d.marker_size, d.max_correction_bits, d.n_markers, d.bytes_list.ctypes.data)
    od._keep = d.bytes_list
    return od


def to_gray(img: np.ndarray, enc: int) -> np.ndarray:
    h, w = img.shape[:2]
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((h, w), dtype=np.uint8)
    rc = lib().ora_to_gray(img.ctypes.data_as(C.c_void_p), w, h, img.strides[0], enc, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def adaptive_threshold(gray: np.ndarray, win: int, c: float) -> np.ndarray:
    g, gp = _u8(gray)
    h, w = g.shape
    out = np.empty_like(g)
    rc = lib().ora_adaptive_threshold(gp, w, h, int(win), C.c_double(c), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def find_contours(mask: np.ndarray):
    """Returns list of (N,2) int32 arrays (x,y) in cv::findContours order, and hole flags."""
    m, mp = _u8(mask)
    h, w = m.shape
    cap_pts = 4 * w * h + 16
    cap_c = w * h + 16
    pts = np.empty((cap_pts, 2), dtype=np.int32)
    off = np.empty(cap_c + 1, dtype=np.int64)
    holes = np.empty(cap_c, dtype=np.int32)
    n = C.c_int(0)
    rc = lib().ora_find_contours(mp, w, h, pts.ctypes.data_as(C.c_void_p), C.c_int64(cap_pts),
                                 off.ctypes.data_as(C.c_void_p), holes.ctypes.data_as(C.c_void_p), cap_c,
                                 C.byref(n))
    assert rc == 0, rc
    k = n.value
    return [pts[off[i]:off[i + 1]].copy() for i in range(k)], holes[:k].copy()


def approx_poly_dp(pts: np.ndarray, eps: float) -> np.ndarray:
    p = np.ascontiguousarray(pts, dtype=np.int32)
    out = np.empty((len(p), 2), dtype=np.int32)
    k = lib().ora_approx_poly_dp(p.ctypes.data_as(C.c_void_p), len(p), C.c_double(eps),
                                 out.ctypes.data_as(C.c_void_p), len(p))
    assert k >= 0
    return out[:k].copy()


def detect(gray: np.ndarray, d, params: OraParams | None = None, cap: int = 1024, trace: bool = False):
    """Returns (ids int32[n], corners float32[n,4,2]) and, if trace, a dict of stage dumps."""
    g, gp = _u8(gray)
    h, w = g.shape
    p = params or default_params()
    od = make_dict(d)
    out = (OraMarker * cap)()
    n = C.c_int(0)
    tr = None
    if trace:
        capc = 65536
        tr = OraTrace()
        ini = (OraCandidate * capc)()
        fil = (OraCandidate * capc)()
        msb = d.marker_size + 2 * p.markerBorderBits
        bits = np.zeros((capc, msb * msb), dtype=np.uint8)
        ident = np.zeros((capc, 2), dtype=np.int32)
        pre = (OraMarker * cap)()
        tr.initial, tr.cap_initial = ini, capc
        tr.filtered, tr.cap_filtered = fil, capc
        tr.bits = bits.ctypes.data_as(C.POINTER(C.c_uint8))
        tr.ident = ident.ctypes.data_as(C.POINTER(C.c_int32))
        tr.presubpix, tr.cap_pre = pre, cap
    rc = lib().ora_detect(gp, w, h, C.byref(p), C.byref(od), out, cap, C.byref(n),
                          C.byref(tr) if tr is not None else None)
    if rc == -4:
        raise CvException("detectMarkers throws on this frame (CORNER_REFINE_CONTOUR: a side of fewer than two points)")
    assert rc == 0, rc
    k = n.value
    ids = np.array([out[i].id for i in range(k)], dtype=np.int32)
    corners = np.array([list(out[i].corners) for i in range(k)], dtype=np.float32).reshape(k, 4, 2)
    if not trace:
        return ids, corners

    def cands(arr, m):
        return dict(
            scale=np.array([arr[i].scale for i in range(m)], dtype=np.int32),
            contour_size=np.array([arr[i].contour_size for i in range(m)], dtype=np.int32),
            start=np.array([[arr[i].start_x, arr[i].start_y] for i in range(m)], dtype=np.int32).reshape(m, 2),
            is_hole=np.array([arr[i].is_hole for i in range(m)], dtype=np.int32),
            corners=np.array([list(arr[i].corners) for i in range(m)], dtype=np.float32).reshape(m, 4, 2),
        )

    t = dict(initial=cands(ini, tr.n_initial), filtered=cands(fil, tr.n_filtered),
             bits=bits[:tr.n_filtered].reshape(tr.n_filtered, msb, msb).copy(), ident=ident[:tr.n_filtered].copy(),
             pre_ids=np.array([pre[i].id for i in range(tr.n_pre)], dtype=np.int32),
             pre_corners=np.array([list(pre[i].corners) for i in range(tr.n_pre)], dtype=np.float32).reshape(tr.n_pre, 4, 2))
    return ids, corners, t


class CvException(Exception):
    """The input on which the reference's OpenCV call raises cv::Exception (the node logs it and publishes nothing)."""


def refine_candidate_lines(contour: np.ndarray, corners: np.ndarray) -> np.ndarray:
    """CORNER_REFINE_CONTOUR on one marker: contour (n, 2) int32 in findContours order, corners (4, 2) float32 -> (4, 2)."""
    c = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
    q = np.ascontiguousarray(corners, dtype=np.float32).reshape(8).copy()
    rc = lib().ora_refine_candidate_lines(c.ctypes.data_as(C.c_void_p), len(c), q.ctypes.data_as(C.c_void_p))
    if rc == -4:
        raise CvException("a side of fewer than two points")
    assert rc == 0, rc
    return q.reshape(4, 2)


def corner_subpix(gray, pts, win=5, max_iter=30, eps=0.01):
    g, gp = _u8(gray)
    h, w = g.shape
    p = np.ascontiguousarray(pts, dtype=np.float32).copy()
    rc = lib().ora_corner_subpix(gp, w, h, p.ctypes.data_as(C.c_void_p), p.size // 2, win, max_iter, C.c_double(eps))
    assert rc == 0
    return p


def identify(gray, d, corners, params=None):
    g, gp = _u8(gray)
    h, w = g.shape
    p = params or default_params()
    od = make_dict(d)
    msb = d.marker_size + 2 * p.markerBorderBits
    bits = np.zeros((msb, msb), dtype=np.uint8)
    rot = C.c_int(-1)
    c = np.ascontiguousarray(corners, dtype=np.float32).reshape(8)
    idx = lib().ora_identify(gp, w, h, C.byref(p), C.byref(od), c.ctypes.data_as(C.c_void_p),
                             bits.ctypes.data_as(C.c_void_p), C.byref(rot))
    return idx, rot.value, bits


def solve_pnp_square(K, D, corners, marker_len):
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    Dp = None if D is None else np.ascontiguousarray(D, dtype=np.float64).reshape(5)
    c = np.ascontiguousarray(corners, dtype=np.float32).reshape(8)
    r = np.zeros(3)
    t = np.zeros(3)
    e = C.c_double(0)
    rc = lib().ora_solve_pnp_square(K.ctypes.data_as(C.c_void_p), Dp.ctypes.data_as(C.c_void_p) if Dp is not None else None,
                                    c.ctypes.data_as(C.c_void_p), C.c_double(marker_len),
                                    r.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), C.byref(e))
    assert rc == 0, rc
    return r, t, e.value


def solve_pnp_points(K, D, obj, img):
    """cv::solvePnP(ITERATIVE) restated, double-precision points (n >= 4, coplanar).  Returns (rvec, tvec)."""
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    Dp = None if D is None else np.ascontiguousarray(D, dtype=np.float64).reshape(5)
    o = np.ascontiguousarray(obj, dtype=np.float64).reshape(-1, 3)
    m = np.ascontiguousarray(img, dtype=np.float64).reshape(-1, 2)
    r = np.zeros(3)
    t = np.zeros(3)
    rc = lib().ora_solve_pnp_d(K.ctypes.data_as(C.c_void_p), Dp.ctypes.data_as(C.c_void_p) if Dp is not None else None,
                               o.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), len(o),
                               r.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return r, t


def fiducial_area(corners) -> float:
    c = np.ascontiguousarray(corners, dtype=np.float32).reshape(8)
    return float(lib().ora_fiducial_area(c.ctypes.data_as(C.c_void_p)))
