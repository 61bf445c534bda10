"""The /fiducial_images overlay (aruco_detect.cpp:381-387) on top of the C-ABI: `to_bgr` = cv_bridge::toCvCopy(msg, BGR8),
`draw_detected_markers` = the part of aruco::drawDetectedMarkers that is restated exactly (the four LINE_8 sides of every
marker; include/fid_abi.h says what is not drawn and why).  Host code on both sides: no GPU needed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FidError, FidMarker

FIRST_CORNER_LINE8 = 1


def to_bgr(image: np.ndarray, encoding: str | None = None) -> np.ndarray:
    img = np.ascontiguousarray(image, dtype=np.uint8)
    if encoding is None:
        encoding = "mono8" if img.ndim == 2 else ("bgra8" if img.shape[2] == 4 else "bgr8")
    h, w = img.shape[:2]
    out = np.empty((h, w, 3), dtype=np.uint8)
    rc = _lib.load().fid_to_bgr(img.ctypes.data, w, h, img.strides[0], _lib.ENC[encoding], out.ctypes.data, out.nbytes)
    if rc != _lib.FID_OK:
        raise FidError(rc, "fid_to_bgr")
    return out


def image_to_bgr8(data: np.ndarray, width: int, height: int, step: int, encoding: str, is_bigendian: bool = False) -> np.ndarray:
    """cv_bridge::toCvCopy(msg, "bgr8") of a sensor_msgs/Image given by its fields (data: the message bytes): the five 8-bit
    encodings, mono16 / bgr16 / rgb16 / bgra16 / rgba16 and the four 8-bit Bayer patterns (fid_image_to_bgr8)."""
    buf = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    out = np.empty((height, width, 3), dtype=np.uint8)
    rc = _lib.load().fid_image_to_bgr8(buf.ctypes.data, width, height, step, encoding.encode(), int(bool(is_bigendian)), out.ctypes.data, out.nbytes)
    if rc != _lib.FID_OK:
        raise FidError(rc, f"fid_image_to_bgr8({encoding})")
    return out


def draw_detected_markers(bgr: np.ndarray, corners: np.ndarray, ids: np.ndarray | None = None, flags: int = 0) -> np.ndarray:
    """In place on a (H, W, 3) uint8 BGR image; corners (n, 4, 2) float32.  Returns the image."""
    if bgr.dtype != np.uint8 or bgr.ndim != 3 or bgr.shape[2] != 3 or bgr.strides[2] != 1 or bgr.strides[1] != 3:
        raise ValueError("draw_detected_markers takes a (H, W, 3) uint8 image with packed pixels")
    c = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 8)
    n = len(c)
    mk = (FidMarker * max(n, 1))()
    for i in range(n):
        mk[i].id = int(ids[i]) if ids is not None else 0
        for j in range(8):
            mk[i].corners[j] = float(c[i, j])
    h, w = bgr.shape[:2]
    rc = _lib.load().fid_draw_detected_markers(bgr.ctypes.data, w, h, bgr.strides[0], mk, n, flags)
    if rc != _lib.FID_OK:
        raise FidError(rc, "fid_draw_detected_markers")
    return bgr
