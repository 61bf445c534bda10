"""Frames that arrive as PNG (compressed_image_transport with `format: png`): fid_png_decode, the library's host-side decoder
(zlib inflates; the reference decodes on the CPU too: cv::imdecode in the subscriber plugin in front of imageCallback,
aruco_detect.cpp:332).  Returns what cv::imdecode(IMREAD_COLOR) does -- (h, w, 3) BGR -- or its BGR2GRAY, ready for
ArucoDetector.detect_markers(img, "bgr8" | "mono8")."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FidError, FidPngInfo


def probe(data: bytes) -> dict:
    L = _lib.load()
    info = FidPngInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.fid_png_probe(buf, len(data), C.byref(info))
    if rc != _lib.FID_OK:
        raise FidError(rc, (L.fid_png_last_error() or b"").decode())
    return {k: getattr(info, k) for k, _ in FidPngInfo._fields_}


def decode(data: bytes, encoding: str = "bgr8") -> np.ndarray:
    """PNG file bytes -> uint8 array, (h, w, 3) BGR for "bgr8" (cv::imdecode(IMREAD_COLOR)), (h, w) for "mono8" (its BGR2GRAY)."""
    if encoding not in ("bgr8", "mono8"):
        raise ValueError("encoding must be bgr8 or mono8")
    info = probe(data)
    h, w = info["height"], info["width"]
    out = np.empty((h, w, 3) if encoding == "bgr8" else (h, w), dtype=np.uint8)
    L = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.fid_png_decode(buf, len(data), _lib.ENC[encoding], out.ctypes.data, out.nbytes, None)
    if rc != _lib.FID_OK:
        raise FidError(rc, (L.fid_png_last_error() or b"").decode())
    return out
