"""The /fiducial_images overlay (aruco_detect.cpp:381-387) on top of the C-ABI: `to_bgr` = cv_bridge::toCvCopy(msg, BGR8),
`draw_detected_markers` = the part of aruco::drawDetectedMarkers that is restated exactly (the four LINE_8 sides of every
marker; include/fid_abi.h says what is not drawn and why).  Host code on both sides: no GPU needed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FidError, FidMarker

FIRST_CORNER_LINE8 = 1
_BYTES_PER_PIXEL = {"mono8": 1, "bgr8": 3, "rgb8": 3, "bgra8": 4, "rgba8": 4, "mono16": 2, "bgr16": 6, "rgb16": 6, "bgra16": 8, "rgba16": 8,
                    "yuv422": 2, "bayer_rggb8": 1, "bayer_bggr8": 1, "bayer_gbrg8": 1, "bayer_grbg8": 1}


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
    # the message BYTES: an ndarray of another dtype (a mono16 frame as uint16) is reinterpreted, never value-converted
    if isinstance(data, np.ndarray):
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    else:
        buf = np.frombuffer(data, dtype=np.uint8)
    width, height, step = int(width), int(height), int(step)
    bpp = _BYTES_PER_PIXEL.get(encoding)
    if width < 1 or height < 1 or (bpp is not None and step < width * bpp) or buf.size < step * height:
        # (the C side reads step * (height - 1) + a row, the Bayer rows two ahead: a short buffer is refused here, as
        #  FiducialsNode::imageCallback refuses data.size() < step * height)
        raise FidError(_lib.FID_E_INVALID_ARG, f"fid_image_to_bgr8({encoding}): {buf.size} bytes for step {step} x height {height}")
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
