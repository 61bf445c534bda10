#!/usr/bin/env python3
"""JPEG fixtures for the ingest path (compressed image_transport in front of the node): tests/golden/jpeg_cases.npz.
Every case is a small JPEG file written by libjpeg-turbo (Pillow) in one of the layouts the decoder supports -- 4:4:4, 4:2:2,
4:2:0, one component, with and without restart intervals, sizes that are not multiples of the MCU -- together with what
libjpeg-turbo itself decodes from it (RGB -> stored as BGR, what cv::imdecode hands to the node).  Plus the SHA-256 of
libjpeg-turbo's decode of the reference's own JPEG fixtures (fiducial_slam/test/test_images/403.jpg and the CompressedImage
frame of fiducial_slam/test/aruco_images.bag), which are read from /root/reference at test time where it is mounted.
Run in the authoring container (needs Pillow and /root/reference)."""
import hashlib
import io
import os
import struct
import sys

import numpy as np
from PIL import Image, features

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/"


def picture(w, h, seed, noise):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 29.0), 128 + 90 * np.cos(xx / 7.0 + seed), (xx * 3 + yy * 5 + seed) % 256], -1)
    # a few hard edges (marker-like)
    a[h // 4:h // 2, w // 4:w // 2] = 20
    a[h // 3:h // 3 + max(h // 8, 1), w // 3:w // 3 + max(w // 8, 1)] = 235
    if noise:
        a = a + rng.normal(0, noise, a.shape)
    return np.clip(a, 0, 255).astype(np.uint8)


def bag_jpeg(path):
    bag = open(path, "rb").read()
    pos = 0
    while True:
        i = bag.find(b"\xff\xd8\xff", pos)
        if i < 0:
            return None
        n = struct.unpack("<I", bag[i - 4:i])[0]
        if 1000 < n < 5_000_000 and bag[i + n - 2:i + n] == b"\xff\xd9":
            return bag[i:i + n]
        pos = i + 3


def main():
    out = {}
    cases = []
    k = 0
    for (w, h) in [(64, 64), (65, 47), (1, 1), (7, 9), (17, 33), (97, 65), (160, 120)]:
        for sub, gray, q, rst, noise in [(2, False, 80, 0, 0), (1, False, 90, 0, 12), (0, False, 60, 0, 25), (2, True, 80, 0, 8), (2, False, 95, 3, 20),
                                         (1, False, 40, 2, 0)]:
            a = picture(w, h, k, noise)
            im = Image.fromarray(a[..., 1] if gray else a)
            b = io.BytesIO()
            kw = dict(quality=q, subsampling=sub)
            if rst:
                kw["restart_marker_blocks"] = rst
            im.save(b, "JPEG", **kw)
            data = b.getvalue()
            rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
            out[f"jpg_{k}"] = np.frombuffer(data, np.uint8)
            out[f"bgr_{k}"] = np.ascontiguousarray(rgb[..., ::-1])
            cases.append((k, w, h, sub, int(gray), q, rst))
            k += 1
    # one progressive file: must be refused, not mis-decoded
    b = io.BytesIO()
    Image.fromarray(picture(48, 48, 99, 5)).save(b, "JPEG", quality=80, progressive=True)
    out["jpg_progressive"] = np.frombuffer(b.getvalue(), np.uint8)
    out["cases"] = np.array(cases, np.int32)
    ref = {}
    p403 = REF + "fiducial_slam/test/test_images/403.jpg"
    if os.path.exists(p403):
        d = open(p403, "rb").read()
        rgb = np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))
        ref["403.jpg"] = hashlib.sha256(np.ascontiguousarray(rgb[..., ::-1]).tobytes()).hexdigest()
        blob = bag_jpeg(REF + "fiducial_slam/test/aruco_images.bag")
        if blob:
            rgb = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
            ref["aruco_images.bag"] = hashlib.sha256(np.ascontiguousarray(rgb[..., ::-1]).tobytes()).hexdigest()
    out["reference_sha256"] = np.array([f"{k}={v}" for k, v in sorted(ref.items())])
    out["made_with"] = np.array([f"Pillow libjpeg {features.version('jpg')} turbo={features.check_feature('libjpeg_turbo')}"])
    path = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", k, "cases;", ref)


if __name__ == "__main__":
    sys.exit(main())
