"""Host-side mirror of the reference's detector interface, on top of the C-ABI.

`ArucoDetector.detect_markers(image)` has the argument meaning and result shape of
`aruco::detectMarkers(image, dictionary, corners, ids, detectorParams)` as aruco_detect calls it
(aruco_detect/src/aruco_detect.cpp:350): corners are (n, 4, 2) float32 in TL,TR,BR,BL order of the
canonical marker, ids (n,) int32, in OpenCV's output order.  `estimate_pose_single_markers` mirrors
`FiducialsNode::estimatePoseSingleMarkers` (:223-255) plus the per-marker error/area values that
poseEstimateCallback publishes (:480-497).

Everything here runs on the MI355X through libfid_amd.so; there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import FidCandidate, FidDict, FidError, FidLimits, FidMarker, FidParams, FidPoseOut
from .dictionary import Dictionary, get_predefined_dictionary


def default_params() -> FidParams:
    p = FidParams()
    _lib.load().fid_default_params(C.byref(p))
    return p


@dataclass
class PoseResult:
    rvecs: np.ndarray  # (n, 3)
    tvecs: np.ndarray  # (n, 3)
    image_error: np.ndarray  # (n,)
    object_error: np.ndarray
    fiducial_area: np.ndarray


_MARKER_DT = np.dtype([("id", "<i4"), ("corners", "<f4", (8,))])  # fid_marker
_POSE_DT = np.dtype([("rvec", "<f8", (3,)), ("tvec", "<f8", (3,)), ("image_error", "<f8"), ("object_error", "<f8"), ("fiducial_area", "<f8")])


def _poses_view_to_result(v) -> PoseResult:
    """v: a structured view (_POSE_DT) of n fid_pose_out records; copies (the buffer is reused by the next call)."""
    return PoseResult(rvecs=v["rvec"].copy(), tvecs=v["tvec"].copy(), image_error=v["image_error"].copy(),
                      object_error=v["object_error"].copy(), fiducial_area=v["fiducial_area"].copy())


def _poses_to_result(arr, n) -> PoseResult:
    return PoseResult(
        rvecs=np.array([list(arr[i].rvec) for i in range(n)], dtype=np.float64).reshape(n, 3),
        tvecs=np.array([list(arr[i].tvec) for i in range(n)], dtype=np.float64).reshape(n, 3),
        image_error=np.array([arr[i].image_error for i in range(n)]),
        object_error=np.array([arr[i].object_error for i in range(n)]),
        fiducial_area=np.array([arr[i].fiducial_area for i in range(n)]),
    )


class ArucoDetector:
    def __init__(self, dictionary: Dictionary | int | str = 7, params: FidParams | None = None, device: int = 0,
                 max_width: int = 1920, max_height: int = 1080, max_batch: int = 1, max_markers: int = 256,
                 max_candidates: int = 2048, max_starts: int = 0, max_contours: int = 0, max_points: int = 0):
        self._L = _lib.load()
        self.dictionary = dictionary if isinstance(dictionary, Dictionary) else get_predefined_dictionary(dictionary)
        self.params = params or default_params()
        d = self.dictionary
        self._dict_bytes = np.ascontiguousarray(d.bytes_list, dtype=np.uint8)
        fd = FidDict(d.marker_size, d.max_correction_bits, d.n_markers, 0, self._dict_bytes.ctypes.data)
        lim = FidLimits()  # zero = "library default for this context size" (fid_create)
        lim.max_width, lim.max_height, lim.max_batch = max_width, max_height, max_batch
        lim.max_markers_per_frame, lim.max_candidates_per_frame = max_markers, max_candidates
        if max_starts:
            lim.max_starts_per_frame = max_starts
        if max_contours:
            lim.max_contours_per_frame = max_contours
        if max_points:
            lim.max_points_per_frame = max_points
        self.limits = lim
        self._ctx = C.c_void_p()
        rc = self._L.fid_create(C.byref(self.params), C.byref(fd), C.byref(lim), device, C.byref(self._ctx))
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        self.device = device
        self.max_markers = max_markers
        self.max_batch = max_batch
        self._out = (FidMarker * (max_batch * max_markers))()
        self._n = (C.c_int32 * max_batch)()
        self._poses = (FidPoseOut * (max_batch * max_markers))()
        # numpy views of the two result buffers: unpacking twenty markers field by field through ctypes cost the single-frame call
        # ~50 us of Python (round 5; the C-ABI call itself is unchanged)
        assert C.sizeof(FidMarker) == _MARKER_DT.itemsize and C.sizeof(FidPoseOut) == _POSE_DT.itemsize
        self._out_np = np.frombuffer(self._out, dtype=_MARKER_DT)
        self._poses_np = np.frombuffer(self._poses, dtype=_POSE_DT)
        self._last_frames = 0

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._L.fid_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == _lib.FID_E_CV_EXCEPTION:
            raise _lib.CvException(rc, (self._L.fid_last_error(self._ctx) or b"").decode())
        if rc != _lib.FID_OK:
            raise FidError(rc, (self._L.fid_last_error(self._ctx) or b"").decode() or self._L.fid_strerror(rc).decode())

    def set_params(self, params: FidParams):
        """dynamic_reconfigure path (aruco_detect.cpp:257-298)."""
        self._check(self._L.fid_set_params(self._ctx, C.byref(params)))
        self.params = params

    # -- detection ----------------------------------------------------------------------------
    def _unpack(self, nframes):
        res = []
        mm = self.max_markers
        for f in range(nframes):
            n = max(int(self._n[f]), 0)  # (-1: the frame on which the reference's detectMarkers throws)
            v = self._out_np[f * mm:f * mm + n]
            res.append((v["corners"].reshape(n, 4, 2).copy(), v["id"].copy()))
        self._last_frames = nframes
        return res

    def detect_markers(self, image: np.ndarray, encoding: str | None = None):
        """One frame from host memory.  image: (H, W) uint8 mono8, (H, W, 3) bgr8/rgb8 or (H, W, 4) bgra8/rgba8 (default bgr8, the
        encoding imageCallback requests from cv_bridge, :348).  Returns (corners, ids)."""
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim not in (2, 3):
            raise FidError(_lib.FID_E_INVALID_ARG, "image must be uint8 HxW or HxWx3")
        if encoding is None:
            encoding = "mono8" if img.ndim == 2 else ("bgra8" if img.shape[2] == 4 else "bgr8")
        if img.strides[-1] != 1 or (img.ndim == 3 and img.strides[1] != img.shape[2]):
            img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        rc = self._L.fid_detect(self._ctx, img.ctypes.data, w, h, img.strides[0], _lib.ENC[encoding], self._out,
                                self.max_markers, self._n)
        self._check(rc)
        return self._unpack(1)[0]

    def detect_image(self, data, width: int, height: int, step: int, encoding: str, is_bigendian: bool = False):
        """One sensor_msgs/Image by its fields (data = the message bytes), in ANY encoding cv_bridge::toCvCopy(msg, BGR8) converts
        (aruco_detect.cpp:348): the five 8-bit layouts, and since ABI 7 bayer_{rggb,bggr,gbrg,grbg}8, mono16 / bgr16 / rgb16 / bgra16 /
        rgba16 (either byte order) and yuv422 -- the conversion is the device's first kernel, the message bytes are what crosses
        PCIe.  Returns (corners, ids)."""
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1) if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
        if encoding not in _lib.ENC:
            raise FidError(_lib.FID_E_UNSUPPORTED, f"cv_bridge exception: unsupported encoding {encoding}")
        width, height, step = int(width), int(height), int(step)
        if width < 1 or height < 1 or step < width * _lib.ENC_BYTES_PER_PIXEL[encoding] or buf.size < step * height:
            raise FidError(_lib.FID_E_INVALID_ARG, f"image is wrongly formed: {buf.size} bytes for step {step} x height {height} ({encoding})")
        rc = self._L.fid_detect(self._ctx, buf.ctypes.data, width, height, step, _lib.encoding_value(encoding, is_bigendian), self._out,
                                self.max_markers, self._n)
        self._check(rc)
        return self._unpack(1)[0]

    def detect_markers_batch(self, images: np.ndarray, encoding: str | None = None, unpack: bool = True, width: int | None = None):
        """Frames in host memory (pinned or pageable): they go up sub-batch by sub-batch while the call runs.  images: (F, H, W)
        mono8 / Bayer, (F, H, W, C) 8-bit colour, or -- with `width` given -- (F, H, step) message rows of a multi-byte encoding."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        if encoding is None:
            encoding = "mono8" if imgs.ndim == 3 else "bgr8"
        f, h, w = imgs.shape[:3]
        if width is not None:
            w = int(width)
        rc = self._L.fid_detect_batch(self._ctx, imgs.ctypes.data, f, w, h, imgs.strides[1], imgs.strides[0],
                                      _lib.ENC[encoding], self._out, self.max_markers, self._n)
        self._check(rc)
        self._last_frames = f
        if not unpack:
            return [int(self._n[k]) for k in range(f)]
        return self._unpack(f)

    def detect_markers_device(self, data_ptr: int, nframes: int, width: int, height: int, stride: int | None = None,
                              frame_stride: int | None = None, encoding: str = "mono8", unpack: bool = True):
        """Frames already resident in HBM (e.g. a torch uint8 tensor's data_ptr() on this device)."""
        bpp = _lib.ENC_BYTES_PER_PIXEL[encoding]
        stride = stride or width * bpp
        frame_stride = frame_stride or stride * height
        rc = self._L.fid_detect_device(self._ctx, C.c_void_p(data_ptr), nframes, width, height, stride, frame_stride,
                                       _lib.ENC[encoding], self._out, self.max_markers, self._n)
        self._check(rc)
        self._last_frames = nframes
        if not unpack:
            return [int(self._n[f]) for f in range(nframes)]
        return self._unpack(nframes)

    def submit_device(self, data_ptr: int, nframes: int, width: int, height: int, stride: int | None = None,
                      frame_stride: int | None = None, encoding: str = "mono8", after: "ArucoDetector | None" = None) -> None:
        """First half of detect_markers_device: enqueue the batch and return at once (fid_submit_device).  The frames must stay
        where they are until collect().  after = another detector: start when its batch in flight is past its chip-filling
        kernels (fid_order_after)."""
        if after is not None:
            self._check(self._L.fid_order_after(self._ctx, after._ctx))
        bpp = _lib.ENC_BYTES_PER_PIXEL[encoding]
        stride = stride or width * bpp
        frame_stride = frame_stride or stride * height
        self._check(self._L.fid_submit_device(self._ctx, C.c_void_p(data_ptr), nframes, width, height, stride, frame_stride,
                                              _lib.ENC[encoding]))
        self._submitted = nframes

    def submit_batch(self, images: np.ndarray, encoding: str | None = None, after: "ArucoDetector | None" = None) -> None:
        """First half of detect_markers_batch (fid_submit_batch): frames in host memory -- a C-contiguous uint8 array that the
        caller keeps alive and unchanged until collect()."""
        imgs = images
        if imgs.dtype != np.uint8 or not imgs.flags.c_contiguous:
            raise ValueError("submit_batch takes a C-contiguous uint8 array (no hidden copy: the memory is read until collect())")
        if encoding is None:
            encoding = "mono8" if imgs.ndim == 3 else "bgr8"
        if after is not None:
            self._check(self._L.fid_order_after(self._ctx, after._ctx))
        f, h, w = imgs.shape[:3]
        self._check(self._L.fid_submit_batch(self._ctx, imgs.ctypes.data, f, w, h, imgs.strides[1], imgs.strides[0], _lib.ENC[encoding]))
        self._submitted = f
        self._held = imgs  # (keeps the array alive)

    def collect(self, unpack: bool = True):
        """Second half: wait for the submitted batch and return what detect_markers_device would have (fid_collect)."""
        self._check(self._L.fid_collect(self._ctx, self._out, self.max_markers, self._n))
        self._held = None
        nframes = self._last_frames = self._submitted
        if not unpack:
            return [int(self._n[f]) for f in range(nframes)]
        return self._unpack(nframes)

    # -- pose ---------------------------------------------------------------------------------
    def estimate_pose_single_markers(self, corners: np.ndarray, ids: np.ndarray, fiducial_len: float, K, D,
                                     fiducial_len_override: dict | None = None) -> PoseResult:
        corners = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 8)
        n = corners.shape[0]
        mk = (FidMarker * max(n, 1))()
        lens = (C.c_double * max(n, 1))()
        for i in range(n):
            mk[i].id = int(ids[i])
            for j in range(8):
                mk[i].corners[j] = float(corners[i, j])
            lens[i] = float((fiducial_len_override or {}).get(int(ids[i]), fiducial_len))  # :241-244
        Kc = (C.c_double * 9)(*np.asarray(K, dtype=np.float64).reshape(9))
        Dc = (C.c_double * 5)(*np.asarray(D, dtype=np.float64).reshape(-1)[:5])
        out = (FidPoseOut * max(n, 1))()
        self._check(self._L.fid_pose(self._ctx, Kc, Dc, mk, lens, n, float(fiducial_len), out))
        return _poses_to_result(out, n)

    def refine_contour_corners(self, contours, corners) -> np.ndarray:
        """aruco.cpp _refineCandidateLines (CORNER_REFINE_CONTOUR) for markers given by their contours (list of (n_i, 2) int
        arrays in findContours order) and quads ((m, 4, 2), corners on the contours): the device code the pipeline runs after
        `_filterDetectedMarkers` when cornerRefinementMethod = 2.  Raises CvException where the reference throws."""
        cs = [np.ascontiguousarray(c, dtype=np.int32).reshape(-1, 2) for c in contours]
        off = np.zeros(len(cs) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(c) for c in cs])
        pts = np.ascontiguousarray(np.concatenate(cs) if cs else np.zeros((0, 2), np.int32))
        q = np.ascontiguousarray(corners, dtype=np.float32).reshape(len(cs), 8).copy()
        st = np.zeros(max(len(cs), 1), dtype=np.int32)
        self.last_refine_status = st
        self._check(self._L.fid_refine_contour_corners(self._ctx, pts.ctypes.data, off.ctypes.data, len(cs), q.ctypes.data, st.ctypes.data))
        return q.reshape(-1, 4, 2)

    def pose_last(self, fiducial_len: float, K, D, unpack: bool = True):
        """Poses of the markers found by the last detect_* call, computed without the corners leaving HBM."""
        Kc = (C.c_double * 9)(*np.asarray(K, dtype=np.float64).reshape(9))
        Dc = (C.c_double * 5)(*np.asarray(D, dtype=np.float64).reshape(-1)[:5])
        self._check(self._L.fid_pose_last(self._ctx, Kc, Dc, float(fiducial_len), self._poses, self.max_markers))
        if not unpack:
            return None
        res = []
        for f in range(self._last_frames):
            n = max(int(self._n[f]), 0)
            res.append(_poses_view_to_result(self._poses_np[f * self.max_markers:f * self.max_markers + n]))
        return res

    # -- stage taps for parity tests ------------------------------------------------------------
    def tap(self, which: int) -> np.ndarray:
        nbytes = self._L.fid_tap_bytes(self._ctx, which)
        buf = np.zeros(nbytes, dtype=np.uint8)
        self._check(self._L.fid_tap_read(self._ctx, which, buf.ctypes.data, nbytes))
        return buf

    def tap_counts(self):
        return self.tap(_lib.TAP_COUNTS).view(np.int32).reshape(-1, 12)

    def tap_masks(self, nframes, nscales, h, w):
        ww = (w + 31) // 32
        m = self.tap(_lib.TAP_MASKS).view(np.uint32).reshape(nframes, nscales, h, ww)
        bits = np.unpackbits(m.view(np.uint8).reshape(nframes, nscales, h, ww * 4), axis=-1, bitorder="little")
        return bits[..., :w]

    def tap_candidates(self, filtered: bool = False):
        raw = self.tap(_lib.TAP_FILTERED if filtered else _lib.TAP_CANDIDATES)
        dt = np.dtype([("scale", "<i4"), ("contour_size", "<i4"), ("start_x", "<i4"), ("start_y", "<i4"),
                       ("is_hole", "<i4"), ("corners", "<f4", (8,))])
        return raw.view(dt).reshape(-1, self.limits.max_candidates_per_frame)

    def tap_bits(self):
        msb = self.dictionary.marker_size + 2 * self.params.markerBorderBits
        return self.tap(_lib.TAP_BITS).reshape(-1, self.limits.max_candidates_per_frame, msb, msb)

    def tap_ident(self):
        return self.tap(_lib.TAP_IDENT).view(np.int32).reshape(-1, self.limits.max_candidates_per_frame, 2)

    def tap_presubpix(self):
        dt = np.dtype([("id", "<i4"), ("corners", "<f4", (8,))])
        return self.tap(_lib.TAP_PRESUBPIX).view(dt).reshape(-1, self.max_markers)

    def stage_ms(self) -> dict:
        ms = (C.c_float * 32)()
        names = C.POINTER(C.c_char_p)()
        n = self._L.fid_last_stage_ms(self._ctx, ms, 32, C.byref(names))
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def last_launches(self) -> int:
        """Launches per kernel in the last detect call (sub-batches on separate streams)."""
        return int(self._L.fid_last_launches(self._ctx))

    @property
    def stream(self) -> int:
        return int(self._L.fid_stream(self._ctx) or 0)
