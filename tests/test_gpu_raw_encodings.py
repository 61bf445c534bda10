"""ABI 7 (round 6): the raw-camera encodings on the DEVICE.  fid_detect / fid_detect_batch / fid_submit_batch take bayer_*8, mono16 /
bgr16 / rgb16 / bgra16 / rgba16 (both byte orders) and yuv422 as published; cv_bridge::toCvCopy(msg, BGR8) (aruco_detect.cpp:348) and
detectMarkers' BGR2GRAY are one pass of the first kernel (k_raw_to_gray).  The checker is the HOST statement of the same rules that
round 5 shipped as the product (fid_image_to_bgr8, fid_draw.hip) followed by the oracle's BGR2GRAY: the device's gray tap must equal
it byte for byte, at even / odd / smallest sizes and with padded rows; then markers through the node's path."""
import numpy as np
import pytest

import oracle
from fiducials_amd import _lib, overlay
from fiducials_amd._lib import FidError
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.dictionary import get_predefined_dictionary

pytestmark = pytest.mark.gpu

BAYER = ("bayer_rggb8", "bayer_bggr8", "bayer_gbrg8", "bayer_grbg8")
WIDE = ("mono16", "bgr16", "rgb16", "bgra16", "rgba16")
SIZES = ((64, 48), (61, 37), (8, 8), (9, 8), (10, 9), (11, 13), (258, 21), (1023, 9))


@pytest.fixture(scope="module")
def det():
    d = ArucoDetector("DICT_5X5_250", max_width=1920, max_height=1080, max_batch=4)
    yield d
    d.close()


def host_gray(rows, w, h, step, enc, be=False):
    bgr = overlay.image_to_bgr8(rows, w, h, step, enc, is_bigendian=be)
    return oracle.to_gray(bgr, 1)  # (enc 1 = bgr8: OpenCV's RGB2Gray<uchar> fixed point)


def message(rng, w, h, enc, pad):
    bpp = _lib.ENC_BYTES_PER_PIXEL[enc]
    step = w * bpp + pad
    rows = rng.integers(0, 256, (h, step), dtype=np.uint8)
    return rows, step


@pytest.mark.parametrize("enc", BAYER + WIDE + ("yuv422",))
def test_device_gray_equals_the_host_conversion(det, enc):
    rng = np.random.default_rng(hash(enc) & 0xffff)
    for (w, h) in SIZES:
        if enc == "yuv422" and (w & 1):
            continue
        for pad in (0, 5):
            for be in ((False, True) if enc in WIDE else (False,)):
                rows, step = message(rng, w, h, enc, pad)
                if enc in WIDE and w * h <= 4096:
                    # (every 16-bit value near a rounding tie of 255 / 65535 somewhere: steps of 257 are the exact 8-bit levels)
                    v = rows[:, : w * _lib.ENC_BYTES_PER_PIXEL[enc]].view(np.uint16)
                    v[:] = (rng.integers(0, 256, v.shape) * 257 + rng.integers(-129, 130, v.shape)).clip(0, 65535).astype(np.uint16)
                det.detect_image(rows, w, h, step, enc, is_bigendian=be)
                got = det.tap(_lib.TAP_GRAY).reshape(h, w)
                want = host_gray(rows, w, h, step, enc, be)
                assert np.array_equal(got, want), (enc, w, h, pad, be, np.argwhere(got != want)[:4])


def test_all_16_bit_values_and_the_uyvy_cube(det):
    allv = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    for be in (False, True):
        src = allv.byteswap() if be else allv
        det.detect_image(src.view(np.uint8), 256, 256, 512, "mono16", is_bigendian=be)
        assert np.array_equal(det.tap(_lib.TAP_GRAY).reshape(256, 256), host_gray(src.view(np.uint8), 256, 256, 512, "mono16", be))
    # every (Y, U, V) on a coarse lattice + the extremes
    ys = np.array([0, 1, 15, 16, 17, 64, 128, 200, 234, 235, 236, 254, 255], np.uint8)
    cs = np.array([0, 1, 16, 64, 127, 128, 129, 192, 240, 254, 255], np.uint8)
    quads = np.array([(u, y0, v, y1) for u in cs for v in cs for y0 in ys for y1 in ys[::3]], np.uint8)
    n = len(quads) // 16 * 16
    img = quads[:n].reshape(-1, 16 * 4)  # rows of 32 pixels
    h, w = img.shape[0], 32
    det.detect_image(img, w, h, 64, "yuv422")
    assert np.array_equal(det.tap(_lib.TAP_GRAY).reshape(h, w), host_gray(img, w, h, 64, "yuv422"))


def test_markers_from_a_bayer_mosaic_and_a_mono16_frame(det):
    """The node's path: a colour scene seen through each Bayer pattern -> the markers fid_detect finds on the mosaic are the
    markers it finds on the host-made BGR8 copy (round 5's road), corners bit for bit; a mono16 frame of v * 257 gives the 8-bit
    frame's markers; the batch entry points take the encodings too."""
    from fiducials_amd.synth import make_frame

    d = get_predefined_dictionary("DICT_5X5_250")
    fr = make_frame(d, seed=11, width=640, height=480, n_markers=5, side_range=(70, 120))
    gray = fr.image
    c0, i0 = det.detect_markers(gray)
    assert len(i0) == 5
    # a tinted scene: channels differ, so the demosaicing matters
    scene = np.stack([(gray.astype(np.int32) * 7 // 8), gray.astype(np.int32), (gray.astype(np.int32) * 3 // 4 + 30)], -1).clip(0, 255).astype(np.uint8)  # B, G, R
    for enc, (r0, b0) in zip(BAYER, (((0, 0), (1, 1)), ((1, 1), (0, 0)), ((1, 0), (0, 1)), ((0, 1), (1, 0)))):
        mosaic = scene[..., 1].copy()
        mosaic[r0[0]::2, r0[1]::2] = scene[r0[0]::2, r0[1]::2, 2]
        mosaic[b0[0]::2, b0[1]::2] = scene[b0[0]::2, b0[1]::2, 0]
        c_dev, i_dev = det.detect_image(mosaic, 640, 480, 640, enc)
        bgr = overlay.image_to_bgr8(mosaic, 640, 480, 640, enc)
        c_host, i_host = det.detect_markers(bgr, encoding="bgr8")
        assert i_dev.tolist() == i_host.tolist() and len(i_dev) == 5, enc
        assert np.array_equal(c_dev, c_host), enc
        assert sorted(i_dev.tolist()) == sorted(i0.tolist())
    wide = (gray.astype(np.uint16) * 257)
    for be in (False, True):
        src = wide.byteswap() if be else wide
        c16, i16 = det.detect_image(src, 640, 480, 1280, "mono16", is_bigendian=be)
        assert i16.tolist() == i0.tolist() and np.array_equal(c16, c0)
    # batches: two mosaics through fid_detect_batch, a stream through fid_submit_batch / fid_collect
    two = np.stack([gray, gray[::-1].copy()])
    res = det.detect_markers_batch(two, encoding="bayer_grbg8")
    single = [det.detect_image(two[k], 640, 480, 640, "bayer_grbg8") for k in range(2)]
    for k in range(2):
        assert res[k][1].tolist() == single[k][1].tolist() and np.array_equal(res[k][0], single[k][0])
    det.submit_batch(two, encoding="bayer_grbg8")
    res2 = det.collect()
    for k in range(2):
        assert res2[k][1].tolist() == single[k][1].tolist() and np.array_equal(res2[k][0], single[k][0])


def test_refusals(det):
    z = np.zeros((16, 64), np.uint8)
    with pytest.raises(FidError) as e:
        det.detect_image(z, 15, 16, 64, "yuv422")  # odd width
    assert e.value.status == _lib.FID_E_INVALID_ARG
    with pytest.raises(FidError) as e:
        det.detect_image(z, 40, 16, 64, "mono16")  # a step smaller than a row
    assert e.value.status == _lib.FID_E_INVALID_ARG
    with pytest.raises(FidError) as e:
        det.detect_image(z, 16, 16, 64, "bayer_rggb16")
    assert e.value.status == _lib.FID_E_UNSUPPORTED
    L = _lib.load()
    import ctypes as C
    n = C.c_int32(0)
    out = (_lib.FidMarker * 4)()
    # FID_ENC_BIGENDIAN on an 8-bit encoding, an unknown value
    for bad in (0x100 | 5, 0x100, 15, 0x200 | 9):
        assert L.fid_detect(det._ctx, z.ctypes.data, 16, 16, 64, bad, out, 4, C.byref(n)) == _lib.FID_E_INVALID_ARG
    enc, bpp = C.c_int(0), C.c_int32(0)
    assert L.fid_encoding_from_string(b"rgba16", 1, C.byref(enc), C.byref(bpp)) == _lib.FID_OK and enc.value == (0x100 | 13) and bpp.value == 8
    assert L.fid_encoding_from_string(b"bayer_gbrg8", 1, C.byref(enc), C.byref(bpp)) == _lib.FID_OK and enc.value == 7 and bpp.value == 1
    assert L.fid_encoding_from_string(b"32FC1", 0, C.byref(enc), None) == _lib.FID_E_UNSUPPORTED
