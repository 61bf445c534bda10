"""The JPEG oracle (oracle/jpeg_oracle.c: baseline decode as cv::imdecode / libjpeg-turbo does it in front of the node when
image_transport runs compressed, aruco_detect.launch:6) against libjpeg-turbo's own output: the committed fixtures
(tests/golden/jpeg_cases.npz, made by tools/make_jpeg_golden.py with Pillow), libjpeg-turbo itself where Pillow is importable,
and the reference's own JPEG files where /root/reference is mounted."""
import hashlib
import io
import os
import struct

import numpy as np
import pytest

from oracle import jpeg as oj

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")
REF = "/root/reference/"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_oracle_equals_libjpeg_turbo_on_the_fixtures(gold):
    for k, w, h, sub, gray, q, rst in gold["cases"].tolist():
        data = gold[f"jpg_{k}"].tobytes()
        i = oj.info(data)
        assert (i["width"], i["height"]) == (w, h) and i["restart"] == rst and i["ncomp"] == (1 if gray else 3)
        assert np.array_equal(oj.decode(data), gold[f"bgr_{k}"]), (k, w, h, sub, gray, q, rst)


def test_unsupported_and_broken_files_are_refused(gold):
    with pytest.raises(oj.JpegError) as e:
        oj.decode(gold["jpg_progressive"].tobytes())
    assert e.value.status == -2
    with pytest.raises(oj.JpegError):
        oj.decode(b"not a jpeg at all")
    data = gold["jpg_0"].tobytes()
    with pytest.raises(oj.JpegError):
        oj.info(data[:40])


def test_stage_outputs_are_consistent(gold):
    """coefficients and planes come in the layout the GPU tests compare against: component after component, MCU padded"""
    data = gold["jpg_5"].tobytes()
    i = oj.info(data)
    bgr, coefs, planes = oj.decode(data, stages=True)
    nb = i["bw0"] * i["bh0"] + 2 * i["bw1"] * i["bh1"]
    assert coefs.size == nb * 64 and planes.size == nb * 64
    y = planes[:i["bw0"] * i["bh0"] * 64].reshape(i["bh0"] * 8, i["bw0"] * 8)
    assert y.shape[0] >= i["height"] and y.shape[1] >= i["width"]
    assert np.abs(coefs).max() > 0


def test_against_libjpeg_turbo_directly():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    n = 0
    for (w, h) in [(33, 17), (80, 48), (129, 77)]:
        for sub in (0, 1, 2):
            for q in (25, 75, 98):
                for rst in (0, 2):
                    a = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
                    a[h // 4:h // 2] = a[h // 4:h // 2] // 8
                    b = io.BytesIO()
                    kw = dict(quality=q, subsampling=sub)
                    if rst:
                        kw["restart_marker_blocks"] = rst
                    PIL.fromarray(a).save(b, "JPEG", **kw)
                    data = b.getvalue()
                    ref = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
                    assert np.array_equal(oj.decode(data)[..., ::-1], ref), (w, h, sub, q, rst)
                    n += 1
    assert n == 54


def _bag_jpeg(path):
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


@pytest.mark.skipif(not os.path.exists(REF + "fiducial_slam/test/test_images/403.jpg"), reason="needs /root/reference")
def test_the_references_own_jpeg_files(gold):
    """fiducial_slam/test/test_images/403.jpg (auto_init_403_test) and the CompressedImage frame of aruco_images.bag: the
    oracle's decode has the SHA-256 libjpeg-turbo's decode had when the fixtures were made."""
    want = dict(s.split("=") for s in gold["reference_sha256"].tolist())
    d = open(REF + "fiducial_slam/test/test_images/403.jpg", "rb").read()
    assert hashlib.sha256(oj.decode(d).tobytes()).hexdigest() == want["403.jpg"]
    blob = _bag_jpeg(REF + "fiducial_slam/test/aruco_images.bag")
    assert blob is not None
    assert hashlib.sha256(oj.decode(blob).tobytes()).hexdigest() == want["aruco_images.bag"]


def test_product_header_parse_agrees_with_the_oracle(gold):
    """fid_jpeg_probe is host-only code of the product library (no device needed): on every fixture it reports what the
    oracle's parser reports, and it refuses what the decoder does not support."""
    from fiducials_amd import jpeg as fj
    from fiducials_amd._lib import FidError

    for k, w, h, sub, gray, q, rst in gold["cases"].tolist():
        data = gold[f"jpg_{k}"].tobytes()
        a, b = fj.probe(data), oj.info(data)
        assert (a["width"], a["height"], a["components"], a["h_samp"], a["v_samp"], a["restart_interval"]) == \
               (b["width"], b["height"], b["ncomp"], b["hmax"], b["vmax"], b["restart"])
        assert a["blocks_w"][0] == b["bw0"] and a["blocks_h"][0] == b["bh0"]
        if not gray:
            assert a["blocks_w"][1] == b["bw1"] and a["blocks_h"][1] == b["bh1"]
        assert 0 < a["scan_bytes"] < len(data)
    with pytest.raises(FidError) as e:
        fj.probe(gold["jpg_progressive"].tobytes())
    assert e.value.status == 6  # FID_E_UNSUPPORTED
    with pytest.raises(FidError):
        fj.probe(b"\xff\xd8\xff\xe0 short")
    # three components that are NOT YCbCr (libjpeg / cv::imdecode skip the colour conversion for them, the device always converts):
    # refused, never another image.  (a) component ids 'R', 'G', 'B' in SOF0 and SOS; (b) an APP14 "Adobe" segment with transform 0
    color = next(gold[f"jpg_{k}"].tobytes() for k, w, h, sub, gray, q, rst in gold["cases"].tolist() if not gray)
    sof = color.index(b"\xff\xc0")
    sos = color.index(b"\xff\xda")
    ids = [color[sof + 10 + 3 * c] for c in range(3)]
    rgb = bytearray(color)
    for c, ch in enumerate(b"RGB"):
        assert rgb[sof + 10 + 3 * c] == ids[c] and rgb[sos + 5 + 2 * c] == ids[c]
        rgb[sof + 10 + 3 * c] = ch
        rgb[sos + 5 + 2 * c] = ch
    with pytest.raises(FidError) as e:
        fj.probe(bytes(rgb))
    assert e.value.status == 6
    adobe = color[:2] + b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00" + color[2:]
    with pytest.raises(FidError) as e:
        fj.probe(adobe)
    assert e.value.status == 6
    ycck = color[:2] + b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x01" + color[2:]  # transform 1 = YCbCr: fine
    assert fj.probe(ycck)["components"] == 3
