"""JPEG ingest on the device -- Python mirror of the `fid_jpeg_*` entry points (include/fid_abi.h): what image_transport's
compressed subscriber + cv::imdecode do in front of FiducialsNode::imageCallback when the node runs with its launch default
`transport:=compressed` (aruco_detect/launch/aruco_detect.launch:6)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FidError

TAP_COEFS, TAP_PLANES = 0, 1


def probe(data: bytes) -> dict:
    """Header parse on the host: size, components, sampling, restart interval, block counts."""
    L = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    info = _lib.FidJpegInfo()
    rc = L.fid_jpeg_probe(C.cast(buf, C.c_void_p), len(data), C.byref(info))
    if rc != _lib.FID_OK:
        raise FidError(rc, L.fid_strerror(rc).decode())
    return {"width": info.width, "height": info.height, "components": info.components, "h_samp": info.h_samp, "v_samp": info.v_samp,
            "restart_interval": info.restart_interval, "blocks_w": list(info.blocks_w), "blocks_h": list(info.blocks_h),
            "scan_bytes": info.scan_bytes}


class JpegDecoder:
    def __init__(self, max_width: int = 1920, max_height: int = 1080, max_batch: int = 1, device: int = 0):
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._L.fid_jpeg_create(device, max_width, max_height, max_batch, C.byref(self._ctx))
        if rc != _lib.FID_OK:
            raise FidError(rc, self._L.fid_strerror(rc).decode())
        self.max_batch = max_batch
        self.max_width, self.max_height = max_width, max_height

    def close(self):
        if self._ctx:
            self._L.fid_jpeg_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc):
        if rc != _lib.FID_OK:
            raise FidError(rc, (self._L.fid_jpeg_last_error(self._ctx) or b"").decode() or self._L.fid_strerror(rc).decode())

    def decode(self, files, encoding: str = "bgr8", to_host: bool = True):
        """files: a bytes object or a list of them (one image size per call).  -> (n, H, W, 3) bgr8 as cv::imdecode returns it,
        or (n, H, W) mono8 = cvtColor(BGR2GRAY) of that; None with to_host=False (the result stays on the device: device_ptr())."""
        single = isinstance(files, (bytes, bytearray, memoryview))
        blobs = [bytes(files)] if single else [f if isinstance(f, bytes) else bytes(f) for f in files]
        n = len(blobs)
        ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(b), C.c_void_p) for b in blobs])  # (the bytes objects themselves: no copy)
        sizes = (C.c_int64 * n)(*[len(b) for b in blobs])
        i = probe(blobs[0])
        bpp = 1 if encoding == "mono8" else 3
        out = np.empty((n, i["height"], i["width"]) + ((3,) if bpp == 3 else ()), np.uint8) if to_host else None
        rc = self._L.fid_jpeg_decode(self._ctx, ptrs, sizes, n, _lib.ENC[encoding], out.ctypes.data if to_host else None,
                                     i["height"] * i["width"] * bpp)
        self._check(rc)
        self._last = (n, i)
        if not to_host:
            return None
        return out[0] if single else out

    def device_ptr(self):
        """-> (device pointer, width, height, stride, frame stride) of the last decode's output"""
        w, h, s, fs = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        p = self._L.fid_jpeg_device_ptr(self._ctx, C.byref(w), C.byref(h), C.byref(s), C.byref(fs))
        return p, w.value, h.value, s.value, fs.value

    def tap(self, which: int, frame: int = 0) -> np.ndarray:
        nb = self._L.fid_jpeg_tap_bytes(self._ctx, which, frame)
        out = np.empty(nb // 2 if which == TAP_COEFS else nb, np.int16 if which == TAP_COEFS else np.uint8)
        self._check(self._L.fid_jpeg_tap_read(self._ctx, which, frame, out.ctypes.data, nb))
        return out

    def last_rounds(self) -> int:
        return int(self._L.fid_jpeg_last_rounds(self._ctx))
