"""JPEG ingest on the device (fid_jpeg_*, fiducials_amd/csrc/fid_jpeg.hip) against the oracle (oracle/jpeg_oracle.c, pinned on
libjpeg-turbo's output): every stage bit for bit -- quantised coefficients (entropy decoding by self-synchronising
sub-sequences + DC prediction), IDCT planes, the BGR image cv::imdecode returns, and the gray image the detector works on."""
import io
import os

import numpy as np
import pytest

from fiducials_amd import jpeg as fj
from fiducials_amd._lib import FidError
from oracle import jpeg as oj

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def gray_of(bgr):
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def check(dec, data):
    bgr_o, coefs_o, planes_o = oj.decode(data, stages=True)
    got = dec.decode(data, "bgr8")
    assert np.array_equal(dec.tap(fj.TAP_COEFS), coefs_o), "coefficients"
    assert np.array_equal(dec.tap(fj.TAP_PLANES), planes_o), "planes"
    assert np.array_equal(got, bgr_o), "bgr"
    assert np.array_equal(dec.decode(data, "mono8"), gray_of(bgr_o)), "gray"


def test_every_fixture_every_stage():
    gold = np.load(GOLD)
    dec = fj.JpegDecoder(max_width=256, max_height=256)
    try:
        for k, w, h, sub, gray, q, rst in gold["cases"].tolist():
            data = gold[f"jpg_{k}"].tobytes()
            i = fj.probe(data)
            assert (i["width"], i["height"], i["restart_interval"]) == (w, h, rst)
            check(dec, data)
            assert np.array_equal(dec.decode(data, "bgr8"), gold[f"bgr_{k}"])  # libjpeg-turbo's own output
        with pytest.raises(FidError) as e:
            dec.decode(gold["jpg_progressive"].tobytes())
        assert e.value.status == 6  # FID_E_UNSUPPORTED, never a wrong image
        with pytest.raises(FidError):
            dec.decode(b"\\xff\\xd8 nothing")
    finally:
        dec.close()
