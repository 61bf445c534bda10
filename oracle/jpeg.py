"""ctypes wrapper of oracle/libjpeg_oracle.so (jpeg_oracle.c).  TEST INFRASTRUCTURE ONLY, like the rest of oracle/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(os.environ.get("ORACLE_SO_DIR") or _HERE, "libjpeg_oracle.so")  # ORACLE_SO_DIR: sanitizer builds
    src = os.path.join(_HERE, "jpeg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libjpeg_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.jpeg_oracle_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]
        L.jpeg_oracle_info.restype = C.c_int
        L.jpeg_oracle_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.jpeg_oracle_decode.restype = C.c_int
        _LIB = L
    return _LIB


class JpegError(Exception):
    def __init__(self, status):
        super().__init__(f"jpeg oracle status {status}")
        self.status = status


def info(data: bytes) -> dict:
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    out = (C.c_int32 * 10)()
    rc = _lib().jpeg_oracle_info(buf, len(data), out)
    if rc != 0:
        raise JpegError(rc)
    keys = ("width", "height", "ncomp", "hmax", "vmax", "bw0", "bh0", "bw1", "bh1", "restart")
    return dict(zip(keys, [int(v) for v in out]))


def decode(data: bytes, stages: bool = False):
    """-> bgr (H, W, 3) uint8 as cv::imdecode(IMREAD_COLOR) gives it; with stages=True also (coefs, planes): flat int16 /
    uint8 arrays, component after component (jpeg_oracle.c:jpeg_oracle_decode)."""
    i = info(data)
    nb = i["bw0"] * i["bh0"] + (2 * i["bw1"] * i["bh1"] if i["ncomp"] == 3 else 0)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    bgr = np.empty((i["height"], i["width"], 3), np.uint8)
    coefs = np.empty(nb * 64, np.int16) if stages else None
    planes = np.empty(nb * 64, np.uint8) if stages else None
    rc = _lib().jpeg_oracle_decode(buf, len(data), coefs.ctypes.data if stages else None, planes.ctypes.data if stages else None,
                                   bgr.ctypes.data)
    if rc != 0:
        raise JpegError(rc)
    return (bgr, coefs, planes) if stages else bgr
