"""Host-side mirror of the reference's STag detector interface (stag_detect/include/stag/Stag.h:41-45), on top of the
C-ABI.  Under construction: `StagDetector(libraryHD, errorCorrection)` mirrors `Stag::Stag`; what exists on the MI355X so
far is the EDPF edge-detection front end of `Stag::detectMarkers` (smoothing, Prewitt gradient map, anchors, anchor
sort); the stages behind it are next.  No CPU path in this package."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FidError

(TAP_SMOOTH, TAP_GRAD, TAP_DIR, TAP_ANCHORS, TAP_SORTED, TAP_EDGEIMG, TAP_SEGMENTS, TAP_SEGPIX, TAP_SMOOTH2, TAP_VGRAD, TAP_VPROB,
 TAP_VSEGMENTS, TAP_LINES, TAP_VLINES, TAP_QUADS, TAP_MARKERS) = range(16)

MARKER_DTYPE = np.dtype([("id", "i4"), ("shift", "i4"), ("corners", "f8", (4, 2)), ("center", "f8", (2,)), ("H", "f8", (3, 3)),
                         ("lineInf", "f8", (3,)), ("projectiveDistortion", "f8"), ("code", "u8")])
POSE_DTYPE = np.dtype([("id", "i4"), ("reserved", "i4"), ("rvec", "f8", (3,)), ("tvec", "f8", (3,)), ("R", "f8", (3, 3))])
QUAD_DTYPE = np.dtype([("corners", "f8", (4, 2)), ("lineInf", "f8", (3,)), ("projectiveDistortion", "f8")])
LINE_DTYPE = np.dtype([("a", "f8"), ("b", "f8"), ("sx", "f8"), ("sy", "f8"), ("ex", "f8"), ("ey", "f8"), ("invert", "i4"),
                       ("segmentNo", "i4"), ("firstPixelIndex", "i4"), ("len", "i4")])


def load_library(hd: int) -> np.ndarray:
    """The codewords of marker library HD<hd> (uint64, four rotations x markers): the published STag tables, extracted by
    tools/make_stag_libraries.py into fiducials_amd/data/stag_HD<hd>.bin (raw little-endian uint64, shared with the C++ host)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"stag_HD{hd}.bin")
    if hd not in (11, 13, 15, 17, 19, 21, 23) or not os.path.exists(path):
        raise FidError(_lib.FID_E_INVALID_ARG, "Invalid library HD. Possible values are 11, 13, 15, 17, 19, 21, or 23")
    return np.ascontiguousarray(np.fromfile(path, dtype="<u8").astype(np.uint64))


class StagDetector:
    def __init__(self, libraryHD: int = 21, errorCorrection: int = 7, max_width: int = 1920, max_height: int = 1080,
                 device: int = 0):
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._L.fid_stag_create(libraryHD, errorCorrection, max_width, max_height, device, C.byref(self._ctx))
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        self.shape = None
        self._words = load_library(libraryHD)  # Decoder::Decoder(libraryHD)
        rc = self._L.fid_stag_load_library(self._ctx, self._words.ctypes.data, len(self._words))
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._L.fid_stag_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def edge_frontend(self, gray: np.ndarray):
        """Runs the EDPF front end on a mono8 image; results stay on the device (read them with tap())."""
        self._run(self._L.fid_stag_edge_frontend, gray)

    def detect_edges(self, gray: np.ndarray):
        """Front end + edge routing (DoDetectEdgesByED): the EdgeMap stays on the device; edge_segments() reads it."""
        self._run(self._L.fid_stag_detect_edges, gray)

    def detect_edges_validated(self, gray: np.ndarray):
        """DetectEdgesByEDPF: detect_edges() + Helmholtz validation; edge_segments(validated=True) reads the result."""
        self._run(self._L.fid_stag_detect_edges_validated, gray)

    def detect_lines(self, gray: np.ndarray):
        """DetectLinesByEDPF up to JoinCollinearLines; lines() reads the result (structured array, LINE_DTYPE)."""
        self._run(self._L.fid_stag_detect_lines, gray)

    def detect_lines_validated(self, gray: np.ndarray):
        """DetectLinesByEDPF complete (EDInterface::runEDPFandEDLines); lines(validated=True) reads EDLines::lines."""
        self._run(self._L.fid_stag_detect_lines_validated, gray)

    def detect_quads(self, gray: np.ndarray):
        """QuadDetector::detectQuads; quads() reads the result (structured array, QUAD_DTYPE)."""
        self._run(self._L.fid_stag_detect_quads, gray)

    def detect_markers_unrefined(self, gray: np.ndarray):
        """Stag::detectMarkers without the final pose refinement; markers() reads the result (MARKER_DTYPE)."""
        self._run(self._L.fid_stag_detect_markers_unrefined, gray)

    def detect_markers(self, gray: np.ndarray) -> np.ndarray:
        """Stag::detectMarkers + getMarkerList(): the markers of a mono8 image (MARKER_DTYPE), pose-refined."""
        img = np.asarray(gray)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise FidError(_lib.FID_E_INVALID_ARG, "image must be uint8 HxW")
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        h, w = img.shape
        n = C.c_int32(0)
        # the markers come back in the caller's buffer (the call's own hand-over) -- reading them through the stage tap afterwards
        # was a second, synchronous device -> host copy per frame
        if getattr(self, "_mbuf", None) is None:
            self._mbuf = np.zeros(512, MARKER_DTYPE)
        rc = self._L.fid_stag_detect_markers(self._ctx, img.ctypes.data, w, h, img.strides[0], self._mbuf.ctypes.data, len(self._mbuf), C.byref(n))
        self.shape = (h, w)
        if rc == _lib.FID_E_CAPACITY:
            m = self.markers()  # more than 512 markers: the tap holds them all; a frame refused in the routing leaves no stage readable
            if len(m) > len(self._mbuf):
                return m
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        return self._mbuf[:n.value].copy()

    def pose_last(self, K, D, marker_size: float) -> np.ndarray:
        """Common::solvePnpSingle for the markers of the last detect_markers*() call (POSE_DTYPE)."""
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        Dv = np.zeros(5) if D is None else np.ascontiguousarray(D, dtype=np.float64).reshape(-1)[:5].copy()
        out = np.zeros(4096, POSE_DTYPE)
        n = C.c_int32(0)
        rc = self._L.fid_stag_pose_last(self._ctx, K.ctypes.data, Dv.ctypes.data, float(marker_size), out.ctypes.data, len(out), C.byref(n))
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        return out[:n.value].copy()

    def queue_stats(self) -> tuple[int, int]:
        """(frames enqueued ahead of their own counts, how many of them had to be run again on the counted road)."""
        q, r = C.c_int32(0), C.c_int32(0)
        self._L.fid_stag_queue_stats(self._ctx, C.byref(q), C.byref(r))
        return q.value, r.value

    def markers(self) -> np.ndarray:
        return self.tap(TAP_MARKERS)

    def quads(self) -> np.ndarray:
        return self.tap(TAP_QUADS)

    def lines(self, validated: bool = False) -> np.ndarray:
        return self.tap(TAP_VLINES if validated else TAP_LINES)

    def edge_segments(self, validated: bool = False):
        """List of (n_i, 2) int32 arrays of (r, c): EdgeMap::segments after detect_edges() / detect_edges_validated()."""
        segs = self.tap(TAP_VSEGMENTS if validated else TAP_SEGMENTS).reshape(-1, 2)
        pix = self.tap(TAP_SEGPIX).reshape(-1, 2)
        return [pix[a:a + n] for a, n in segs]

    def _run(self, fn, gray):
        img = np.asarray(gray)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise FidError(_lib.FID_E_INVALID_ARG, "image must be uint8 HxW")
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        h, w = img.shape
        rc = fn(self._ctx, img.ctypes.data, w, h, img.strides[0])
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        self.shape = (h, w)

    def tap(self, which: int) -> np.ndarray:
        n = self._L.fid_stag_tap_bytes(self._ctx, which)
        buf = np.zeros(max(n, 0), np.uint8)
        if n:
            rc = self._L.fid_stag_tap_read(self._ctx, which, buf.ctypes.data, n)
            if rc != _lib.FID_OK:
                raise FidError(rc, self._L.fid_strerror(rc).decode())
        h, w = self.shape
        if which in (TAP_SMOOTH, TAP_DIR, TAP_ANCHORS, TAP_EDGEIMG, TAP_SMOOTH2):
            return buf.reshape(h, w)
        if which in (TAP_GRAD, TAP_VGRAD):
            return buf.view(np.int16).reshape(h, w)
        if which == TAP_VPROB:
            return buf.view(np.float64)
        if which == TAP_MARKERS:
            return buf.view(MARKER_DTYPE)
        if which == TAP_QUADS:
            return buf.view(QUAD_DTYPE)
        if which in (TAP_LINES, TAP_VLINES):
            return buf.view(LINE_DTYPE)
        return buf.view(np.int32)


class StagPool:
    """Throughput mode (fid_stag_detect_markers_batch): `n_contexts` frame slots.  The frames are a GRID DIMENSION -- the slots
    are cut into groups (32 slots where the pool has at least 64, else two groups; FID_STAG_GROUP overrides), a group carries its
    frames through the pipeline in lockstep on one stream, every kernel launched once per group (fid_stag_batch.h: frame 0's
    arguments + 32 bytes per further frame, round 6), a host thread per group.  Nothing depends on how many hardware queues the HIP
    runtime was given.  A slot is ~0.66 GB for 1080p; 256 slots: 7.9 - 8.2 k frames/s, 32 slots: ~3.5 k.
    detect_markers_batch(frames[F, H, W]) -> (markers per frame, poses per frame)."""

    def __init__(self, libraryHD: int = 21, errorCorrection: int = 7, n_contexts: int = 32, max_width: int = 1920, max_height: int = 1080,
                 device: int = 0):
        self.dets = [StagDetector(libraryHD, errorCorrection, max_width, max_height, device) for _ in range(n_contexts)]
        self._L = self.dets[0]._L
        self._arr = (C.c_void_p * n_contexts)(*[d._ctx.value for d in self.dets])

    def close(self):
        for d in self.dets:
            d.close()

    def detect_markers_batch(self, frames: np.ndarray, K=None, D=None, marker_size: float = 0.18, cap_per_frame: int = 64):
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        F, h, w = fr.shape
        markers = np.zeros((F, cap_per_frame), MARKER_DTYPE)
        poses = np.zeros((F, cap_per_frame), POSE_DTYPE)
        counts = np.zeros(F, np.int32)
        Kp = None if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        Dp = np.zeros(5) if D is None else np.ascontiguousarray(D, dtype=np.float64).reshape(-1)[:5].copy()
        rc = self._L.fid_stag_detect_markers_batch(self._arr, len(self.dets), fr.ctypes.data, F, w, h, w, w * h,
                                                   None if Kp is None else Kp.ctypes.data, Dp.ctypes.data, float(marker_size),
                                                   markers.ctypes.data, poses.ctypes.data, cap_per_frame, counts.ctypes.data)
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        return [markers[f, :counts[f]] for f in range(F)], [poses[f, :counts[f]] for f in range(F)]
