"""ctypes binding of the C-ABI (include/fid_abi.h).  Loading fails loudly if the HIP library is missing:
there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build


class FidParams(C.Structure):
    _fields_ = [
        ("adaptiveThreshConstant", C.c_double),
        ("adaptiveThreshWinSizeMin", C.c_int32),
        ("adaptiveThreshWinSizeMax", C.c_int32),
        ("adaptiveThreshWinSizeStep", C.c_int32),
        ("cornerRefinementMethod", C.c_int32),
        ("cornerRefinementWinSize", C.c_int32),
        ("cornerRefinementMaxIterations", C.c_int32),
        ("cornerRefinementMinAccuracy", C.c_double),
        ("errorCorrectionRate", C.c_double),
        ("minCornerDistanceRate", C.c_double),
        ("markerBorderBits", C.c_int32),
        ("minDistanceToBorder", C.c_int32),
        ("maxErroneousBitsInBorderRate", C.c_double),
        ("minMarkerDistanceRate", C.c_double),
        ("minMarkerPerimeterRate", C.c_double),
        ("maxMarkerPerimeterRate", C.c_double),
        ("minOtsuStdDev", C.c_double),
        ("perspectiveRemoveIgnoredMarginPerCell", C.c_double),
        ("perspectiveRemovePixelPerCell", C.c_int32),
        ("reserved0", C.c_int32),
        ("polygonalApproxAccuracyRate", C.c_double),
    ]


class FidDict(C.Structure):
    _fields_ = [("marker_size", C.c_int32), ("max_correction_bits", C.c_int32), ("n_markers", C.c_int32),
                ("reserved0", C.c_int32), ("bytes", C.c_void_p)]


class FidMarker(C.Structure):
    _fields_ = [("id", C.c_int32), ("corners", C.c_float * 8)]


class FidPoseOut(C.Structure):
    _fields_ = [("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("image_error", C.c_double),
                ("object_error", C.c_double), ("fiducial_area", C.c_double)]


class FidLimits(C.Structure):
    _fields_ = [("max_width", C.c_int32), ("max_height", C.c_int32), ("max_batch", C.c_int32),
                ("max_starts_per_frame", C.c_int32), ("max_contours_per_frame", C.c_int32),
                ("max_candidates_per_frame", C.c_int32), ("max_markers_per_frame", C.c_int32),
                ("max_points_per_frame", C.c_int32)]


class FidCandidate(C.Structure):
    _fields_ = [("scale", C.c_int32), ("contour_size", C.c_int32), ("start_x", C.c_int32), ("start_y", C.c_int32),
                ("is_hole", C.c_int32), ("corners", C.c_float * 8)]


class FidJpegInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32), ("h_samp", C.c_int32), ("v_samp", C.c_int32),
                ("restart_interval", C.c_int32), ("blocks_w", C.c_int32 * 3), ("blocks_h", C.c_int32 * 3), ("scan_bytes", C.c_int64)]


class FidPngInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bit_depth", C.c_int32), ("color_type", C.c_int32), ("interlace", C.c_int32),
                ("gray", C.c_int32)]


FID_OK = 0
FID_E_INVALID_ARG, FID_E_NO_DEVICE, FID_E_HIP, FID_E_CAPACITY, FID_E_OUT_OF_MEMORY, FID_E_UNSUPPORTED, FID_E_CV_EXCEPTION = 1, 2, 3, 4, 5, 6, 7
CORNER_REFINE_NONE, CORNER_REFINE_SUBPIX, CORNER_REFINE_CONTOUR = 0, 1, 2  # aruco::CornerRefineMethod as the node sets it (:700-711)
ENC = {"mono8": 0, "bgr8": 1, "rgb8": 2, "bgra8": 3, "rgba8": 4,
       # ABI 7: what raw camera drivers publish, converted on the device (fid_encoding in include/fid_abi.h)
       "bayer_rggb8": 5, "bayer_bggr8": 6, "bayer_gbrg8": 7, "bayer_grbg8": 8, "mono16": 9, "bgr16": 10, "rgb16": 11, "bgra16": 12, "rgba16": 13,
       "yuv422": 14}
ENC_BIGENDIAN = 0x100  # OR-ed onto a 16-bit encoding (sensor_msgs/Image.is_bigendian)
ENC_BYTES_PER_PIXEL = {"mono8": 1, "bgr8": 3, "rgb8": 3, "bgra8": 4, "rgba8": 4, "bayer_rggb8": 1, "bayer_bggr8": 1, "bayer_gbrg8": 1,
                       "bayer_grbg8": 1, "mono16": 2, "bgr16": 6, "rgb16": 6, "bgra16": 8, "rgba16": 8, "yuv422": 2}


def encoding_value(encoding: str, is_bigendian: bool = False) -> int:
    """fid_encoding of a sensor_msgs/Image encoding string (the table fid_encoding_from_string holds)."""
    v = ENC[encoding]
    return v | ENC_BIGENDIAN if is_bigendian and 9 <= v <= 13 else v


TAP_MASKS, TAP_CANDIDATES, TAP_FILTERED, TAP_BITS, TAP_IDENT, TAP_PRESUBPIX, TAP_COUNTS, TAP_GRAY = range(8)

# every symbol include/fid_abi.h declares
SYMBOLS = [
    "fid_default_params", "fid_default_limits", "fid_create", "fid_destroy", "fid_set_params", "fid_detect",
    "fid_detect_batch", "fid_detect_device", "fid_submit_device", "fid_submit_batch", "fid_collect", "fid_order_after", "fid_pose", "fid_pose_last", "fid_refine_contour_corners", "fid_tap_bytes", "fid_tap_read",
    "fid_last_stage_ms", "fid_last_launches", "fid_stream", "fid_strerror", "fid_last_error", "fid_abi_version",
    "fid_stag_create", "fid_stag_destroy", "fid_stag_edge_frontend", "fid_stag_detect_edges", "fid_stag_detect_edges_validated", "fid_stag_detect_lines", "fid_stag_detect_lines_validated", "fid_stag_detect_quads", "fid_stag_host_tables", "fid_stag_load_library", "fid_stag_detect_markers_unrefined", "fid_stag_detect_markers", "fid_stag_pose_last", "fid_stag_detect_markers_batch", "fid_stag_tap_bytes", "fid_stag_tap_read", "fid_stag_queue_stats",
    "fid_jpeg_probe", "fid_jpeg_create", "fid_jpeg_destroy", "fid_jpeg_decode", "fid_jpeg_device_ptr", "fid_jpeg_tap_bytes", "fid_jpeg_tap_read",
    "fid_jpeg_last_rounds", "fid_jpeg_last_error",
    "fid_png_probe", "fid_png_decode", "fid_png_last_error",
    "fid_to_bgr", "fid_image_to_bgr8", "fid_encoding_from_string", "fid_draw_detected_markers", "fid_dict_load_file", "fid_dict_last_error",
]

_LIB = None


class FidError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"fid status {status}: {msg}")
        self.status = status


class CvException(FidError):
    """FID_E_CV_EXCEPTION: the input on which the reference's OpenCV call throws cv::Exception (the node's imageCallback logs
    it and publishes nothing for the frame, aruco_detect.cpp:391-393)."""


def device_text_sha256(path: str | None = None) -> str | None:
    """sha256 of the gfx950 code object's .text inside the library (ELF -> .hip_fatbin -> clang offload bundle -> the gfx950 ELF ->
    .text), in plain Python.  What the counter files under profiles/ are keyed to since round 5: a rebuild of the same sources
    changes the bundle's metadata (and so the file's hash) but not a byte of device code, and a change on the HOST side of the
    library (fid_draw.hip, the parsers) does not make the device counters stale.  None if the file cannot be read that way."""
    import hashlib
    import struct

    try:
        with open(path or lib_path(), "rb") as fh:
            b = fh.read()

        def sections(e):
            shoff = struct.unpack_from("<Q", e, 0x28)[0]
            es, sn, sx = struct.unpack_from("<HHH", e, 0x3A)
            hs = [struct.unpack_from("<IIQQQQ", e, shoff + i * es) for i in range(sn)]
            so = hs[sx][4]
            return {e[so + h[0]:e.index(b"\0", so + h[0])].decode(): (h[4], h[5]) for h in hs}

        off, size = sections(b)[".hip_fatbin"]
        f = b[off:off + size]
        if f[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
            return None
        n = struct.unpack_from("<Q", f, 24)[0]
        p = 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", f, p)
            p += 24
            triple = f[p:p + tl].decode()
            p += tl
            if "gfx950" in triple:
                co = f[o:o + sz]
                to, ts = sections(co)[".text"]
                return hashlib.sha256(co[to:to + ts]).hexdigest()
    except Exception:  # noqa: BLE001
        return None
    return None


def profile_matches(doc: dict, path: str | None = None) -> bool:
    """Is a counter file under profiles/ (keyed to `device_text_sha256`, older ones to `library_sha256`) about THIS library?"""
    import hashlib

    dev = doc.get("device_text_sha256")
    if dev:
        return dev == device_text_sha256(path)
    try:
        with open(path or lib_path(), "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest() == doc.get("library_sha256")
    except OSError:
        return False


def lib_path() -> str:
    # FID_LIB lets a developer point at an instrumented build (e.g. -DFID_DEBUG_STATS); it is still libfid_amd
    return os.environ.get("FID_LIB") or _build.LIB


def load():
    """Load libfid_amd.so.  Raises if it has not been built (run `python -m fiducials_amd.build` or
    __graft_entry__.build()); never substitutes another implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FidError(FID_E_NO_DEVICE, f"{path} is missing: build the HIP library first (fiducials_amd.build.build())")
    L = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.fid_default_params.argtypes = [C.POINTER(FidParams)]
    L.fid_default_limits.argtypes = [C.POINTER(FidLimits)]
    L.fid_create.argtypes = [C.POINTER(FidParams), C.POINTER(FidDict), C.POINTER(FidLimits), C.c_int, C.POINTER(vp)]
    L.fid_destroy.argtypes = [vp]
    L.fid_destroy.restype = None
    L.fid_set_params.argtypes = [vp, C.POINTER(FidParams)]
    L.fid_detect.argtypes = [vp, vp, i32, i32, i32, C.c_int, C.POINTER(FidMarker), i32, C.POINTER(i32)]
    L.fid_detect_batch.argtypes = [vp, vp, i32, i32, i32, i32, i64, C.c_int, C.POINTER(FidMarker), i32, C.POINTER(i32)]
    L.fid_detect_device.argtypes = [vp, vp, i32, i32, i32, i32, i64, C.c_int, C.POINTER(FidMarker), i32, C.POINTER(i32)]
    L.fid_submit_device.argtypes = [vp, vp, i32, i32, i32, i32, i64, C.c_int]
    L.fid_submit_batch.argtypes = [vp, vp, i32, i32, i32, i32, i64, C.c_int]
    L.fid_collect.argtypes = [vp, C.POINTER(FidMarker), i32, C.POINTER(i32)]
    L.fid_order_after.argtypes = [vp, vp]
    L.fid_pose.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(FidMarker), C.POINTER(C.c_double),
                           i32, C.c_double, C.POINTER(FidPoseOut)]
    L.fid_pose_last.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(FidPoseOut), i32]
    L.fid_refine_contour_corners.argtypes = [vp, vp, vp, i32, vp, vp]
    L.fid_tap_bytes.argtypes = [vp, C.c_int]
    L.fid_tap_bytes.restype = i64
    L.fid_tap_read.argtypes = [vp, C.c_int, vp, i64]
    L.fid_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float), i32, C.POINTER(C.POINTER(C.c_char_p))]
    L.fid_last_stage_ms.restype = i32
    L.fid_last_launches.argtypes = [vp]
    L.fid_last_launches.restype = i32
    L.fid_stream.argtypes = [vp]
    L.fid_stream.restype = vp
    L.fid_strerror.argtypes = [C.c_int]
    L.fid_strerror.restype = C.c_char_p
    L.fid_last_error.argtypes = [vp]
    L.fid_last_error.restype = C.c_char_p
    L.fid_abi_version.restype = i32
    L.fid_stag_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.fid_stag_destroy.argtypes = [vp]
    L.fid_stag_destroy.restype = None
    L.fid_stag_edge_frontend.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_detect_edges.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_detect_edges_validated.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_detect_lines.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_detect_lines_validated.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_detect_quads.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_load_library.argtypes = [vp, vp, i32]
    L.fid_stag_host_tables.argtypes = [i32, i32, vp, i32, vp, vp, vp, vp]
    L.fid_stag_detect_markers_unrefined.argtypes = [vp, vp, i32, i32, i32]
    L.fid_stag_detect_markers.argtypes = [vp, vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
    L.fid_stag_pose_last.argtypes = [vp, vp, vp, C.c_double, vp, i32, C.POINTER(i32)]
    L.fid_stag_detect_markers_batch.argtypes = [vp, i32, vp, i32, i32, i32, i32, i64, vp, vp, C.c_double, vp, vp, i32, vp]
    L.fid_stag_tap_bytes.argtypes = [vp, C.c_int]
    L.fid_stag_tap_bytes.restype = i64
    L.fid_stag_tap_read.argtypes = [vp, C.c_int, vp, i64]
    L.fid_stag_queue_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.fid_jpeg_probe.argtypes = [vp, i64, C.POINTER(FidJpegInfo)]
    L.fid_jpeg_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.fid_jpeg_destroy.argtypes = [vp]
    L.fid_jpeg_destroy.restype = None
    L.fid_jpeg_decode.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), i32, C.c_int, vp, i64]
    L.fid_jpeg_device_ptr.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    L.fid_jpeg_device_ptr.restype = vp
    L.fid_jpeg_tap_bytes.argtypes = [vp, C.c_int, i32]
    L.fid_jpeg_tap_bytes.restype = i64
    L.fid_jpeg_tap_read.argtypes = [vp, C.c_int, i32, vp, i64]
    L.fid_jpeg_last_rounds.argtypes = [vp]
    L.fid_jpeg_last_rounds.restype = i32
    L.fid_jpeg_last_error.argtypes = [vp]
    L.fid_jpeg_last_error.restype = C.c_char_p
    L.fid_png_probe.argtypes = [vp, i64, C.POINTER(FidPngInfo)]
    L.fid_png_decode.argtypes = [vp, i64, C.c_int, vp, i64, C.POINTER(FidPngInfo)]
    L.fid_to_bgr.argtypes = [vp, i32, i32, i32, C.c_int, vp, i64]
    L.fid_image_to_bgr8.argtypes = [vp, i32, i32, i32, C.c_char_p, i32, vp, i64]
    if hasattr(L, "fid_encoding_from_string"):  # (ABI 7; tools/gpu_stag_ab_libs.py also loads the builds of earlier rounds through FID_LIB)
        L.fid_encoding_from_string.argtypes = [C.c_char_p, i32, C.POINTER(C.c_int), C.POINTER(i32)]
    L.fid_draw_detected_markers.argtypes = [vp, i32, i32, i32, C.POINTER(FidMarker), i32, C.c_uint32]
    L.fid_dict_load_file.argtypes = [C.c_char_p, i32, vp, i64, C.POINTER(FidDict)]
    L.fid_dict_last_error.argtypes = []
    L.fid_dict_last_error.restype = C.c_char_p
    L.fid_png_last_error.argtypes = []
    L.fid_png_last_error.restype = C.c_char_p
    _LIB = L
    return L
