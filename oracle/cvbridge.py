"""cv_bridge::toCvCopy(msg, "bgr8") for 16-bit, Bayer and UYVY images: a SECOND STATEMENT of the rules that fid_image_to_bgr8
(fiducials_amd/csrc/fid_draw.hip) and, since round 6, the device kernel k_raw_to_gray implement -- written by the same author from the
same reading of the published OpenCV 4.2 / cv_bridge sources, in another form (whole-array numpy instead of OpenCV's row loops).
It is NOT an independent source: it catches slips of the C statement (indices, parities, rounding), it cannot catch a shared
misreading of OpenCV.  TEST INFRASTRUCTURE ONLY.  The rules: an interior pixel keeps its own colour and takes the other two from the
nearest samples of those colours (two: (a + b + 1) >> 1, four: (a + b + c + d + 2) >> 2); border columns, then border rows, repeat
their neighbours; 16 -> 8 bit is cvRound(float(v) * float(255 / 65535)); UYVY is BT.601 in 20-bit fixed point.
PARITY UNPINNED: OpenCV is not on this machine and the reference holds no fixture in these encodings; one golden vector per encoding
made with real cv2.cvtColor / convertTo would pin it, and is what a deployer with OpenCV should add first."""
from __future__ import annotations

import numpy as np

# colour of the pixel at (row & 1, col & 1); 0 = B, 1 = G, 2 = R
PATTERNS = {"bayer_rggb8": ((2, 1), (1, 0)), "bayer_bggr8": ((0, 1), (1, 2)), "bayer_gbrg8": ((1, 0), (2, 1)), "bayer_grbg8": ((1, 2), (0, 1))}


def bayer_to_bgr(raw: np.ndarray, encoding: str) -> np.ndarray:
    h, w = raw.shape
    out = np.zeros((h, w, 3), np.uint8)
    if h <= 2:
        return out
    pat = PATTERNS[encoding]
    v = raw.astype(np.int32)
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            own = pat[y & 1][x & 1]
            px = [0, 0, 0]
            px[own] = v[y, x]
            if own == 1:
                px[pat[y & 1][(x + 1) & 1]] = (v[y, x - 1] + v[y, x + 1] + 1) >> 1
                px[pat[(y + 1) & 1][x & 1]] = (v[y - 1, x] + v[y + 1, x] + 1) >> 1
            else:
                px[1] = (v[y - 1, x] + v[y + 1, x] + v[y, x - 1] + v[y, x + 1] + 2) >> 2
                px[2 - own] = (v[y - 1, x - 1] + v[y - 1, x + 1] + v[y + 1, x - 1] + v[y + 1, x + 1] + 2) >> 2
            out[y, x] = px
    out[1:h - 1, 0] = out[1:h - 1, 1]
    out[1:h - 1, w - 1] = out[1:h - 1, w - 2]
    out[0] = out[1]
    out[h - 1] = out[h - 2]
    return out


def scale_16_to_8(v: np.ndarray) -> np.ndarray:
    p = v.astype(np.float32) * np.float32(255.0 / 65535.0)
    return np.clip(np.rint(p), 0, 255).astype(np.uint8)  # np.rint: ties to even, like cvRound


def to_bgr8_16bit(img16: np.ndarray, encoding: str) -> np.ndarray:
    """img16: (H, W) or (H, W, C) uint16 in host byte order."""
    v = scale_16_to_8(img16)
    if encoding == "mono16":
        return np.repeat(v[:, :, None], 3, axis=2)
    if encoding in ("bgr16", "bgra16"):
        return np.ascontiguousarray(v[:, :, :3])
    return np.ascontiguousarray(v[:, :, 2::-1])  # rgb16 / rgba16


def uyvy_to_bgr(raw: np.ndarray) -> np.ndarray:
    """raw: (H, W, 2) uint8, bytes U0 Y0 V0 Y1 ... (sensor_msgs 'yuv422'): BT.601 limited range in 20-bit fixed point, as cvtColor
    (COLOR_YUV2BGR_UYVY) states it: channel = clip((max(Y - 16, 0) * 1220542 + 2^19 + c_u (U - 128) + c_v (V - 128)) >> 20)."""
    h, w, _ = raw.shape
    r = raw.astype(np.int64)
    y = np.maximum(r[:, :, 1] - 16, 0) * 1220542
    u = np.repeat(r[:, 0::2, 0] - 128, 2, axis=1)
    v = np.repeat(r[:, 1::2, 0] - 128, 2, axis=1)
    half = 1 << 19
    b = (y + half + 2116026 * u) >> 20
    g = (y + half - 852492 * v - 409993 * u) >> 20
    rr = (y + half + 1673527 * v) >> 20
    return np.clip(np.stack([b, g, rr], axis=2), 0, 255).astype(np.uint8)
