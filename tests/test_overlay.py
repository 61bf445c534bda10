"""The /fiducial_images overlay (aruco_detect.cpp:381-387): fid_to_bgr = cv_bridge::toCvCopy(msg, BGR8),
fid_draw_detected_markers = the four cv::line(LINE_8) sides of aruco::drawDetectedMarkers, against the plain-Python restatement
(oracle/draw.py).  Host code on both sides: runs without a GPU.  What the library does not draw (LINE_AA corner square, id text)
is stated in include/fid_abi.h; nothing here claims it."""
import numpy as np
import pytest

from fiducials_amd import overlay
from fiducials_amd._lib import FidError
from oracle import draw as odraw


def test_to_bgr_every_encoding():
    rng = np.random.default_rng(2)
    g = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(overlay.to_bgr(g), np.repeat(g[:, :, None], 3, axis=2))
    c3 = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(overlay.to_bgr(c3, "bgr8"), c3)
    assert np.array_equal(overlay.to_bgr(c3, "rgb8"), c3[:, :, ::-1])
    c4 = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    assert np.array_equal(overlay.to_bgr(c4, "bgra8"), c4[:, :, :3])
    assert np.array_equal(overlay.to_bgr(c4, "rgba8"), c4[:, :, 2::-1])
    # a strided view (sensor_msgs/Image.step > width * bytes per pixel)
    wide = rng.integers(0, 256, (20, 64), dtype=np.uint8)
    assert np.array_equal(overlay.to_bgr(wide[:, :50])[:, :, 0], wide[:, :50])


def test_line8_restatement_hand_cases():
    # horizontal, vertical, the two diagonals, right-to-left input (drawn left to right), a single point
    assert odraw.line8_pixels(10, 10, (1, 2), (4, 2)) == [(1, 2), (2, 2), (3, 2), (4, 2)]
    assert odraw.line8_pixels(10, 10, (4, 2), (1, 2)) == [(1, 2), (2, 2), (3, 2), (4, 2)]
    assert odraw.line8_pixels(10, 10, (3, 1), (3, 4)) == [(3, 1), (3, 2), (3, 3), (3, 4)]
    assert odraw.line8_pixels(10, 10, (0, 0), (3, 3)) == [(0, 0), (1, 1), (2, 2), (3, 3)]
    assert odraw.line8_pixels(10, 10, (0, 3), (3, 0)) == [(0, 3), (1, 2), (2, 1), (3, 0)]
    assert odraw.line8_pixels(10, 10, (5, 5), (5, 5)) == [(5, 5)]
    # slope 1/3: dx + 1 pixels, one per column, rows change where the error term says
    px = odraw.line8_pixels(20, 20, (0, 0), (9, 3))
    assert len(px) == 10 and [p[0] for p in px] == list(range(10)) and px[-1] == (9, 3)
    assert odraw.line8_pixels(10, 10, (-5, -5), (-1, -2)) == []  # wholly outside


def test_draw_detected_markers_equals_the_restatement():
    rng = np.random.default_rng(8)
    W, H = 640, 480
    quads = []
    for _ in range(60):
        c = rng.uniform([40, 40], [W - 40, H - 40])
        a = rng.uniform(0, 2 * np.pi)
        r = rng.uniform(5, 120)
        ang = a + np.array([0, 0.5, 1.0, 1.5]) * np.pi + rng.uniform(-0.2, 0.2, 4)
        quads.append(np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang)], 1))
    quads = np.array(quads, dtype=np.float32)  # (sub-pixel corners, some of them outside the image: clipLine)
    quads[0] = [[10.5, 10.5], [11.5, 10.5], [11.5, 11.5], [10.5, 11.5]]  # halves round to even
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    got = overlay.draw_detected_markers(base.copy(), quads, np.arange(len(quads)))
    want = odraw.draw_detected_markers(base.copy(), quads)
    assert np.array_equal(got, want), int((got != want).any(axis=2).sum())
    assert (got != base).any()
    # every drawn pixel is the border colour, nothing else changed
    changed = (got != base).any(axis=2)
    assert (got[changed] == (0, 255, 0)).all()
    # the optional first-corner square (LINE_8, not the reference's anti-aliased pixels) only ever adds RED pixels
    # (drawDetectedMarkers: cornerColor = borderColor with val[1] and val[2] swapped = (0, 0, 255) in BGR)
    both = overlay.draw_detected_markers(base.copy(), quads, None, overlay.FIRST_CORNER_LINE8)
    extra = (both != got).any(axis=2)
    assert extra.any() and (both[extra] == (0, 0, 255)).all()


def test_draw_corners_outside_the_int_range():
    """CORNER_REFINE_CONTOUR crosses two fitted lines: near-parallel ones give huge, infinite or NaN corners, and imageCallback
    hands them straight to drawDetectedMarkers.  Point2f -> Point is cvtss2si there (INT_MIN for all of them); the lines are
    clipped in 64-bit arithmetic and nothing is written outside the image (round-4 advisor: a segfault before the fix)."""
    W, H = 320, 240
    bad = [np.inf, -np.inf, np.nan, 1e18, -1e18, 3e9, -3e9, 2147483648.0, -2147483904.0, 2147483520.0]
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (H + 2, W, 3), dtype=np.uint8)  # a guard row above and below the image that is drawn on
    for trial in range(200):
        q = rng.uniform([0, 0], [W, H], (4, 2)).astype(np.float32)
        for _ in range(int(rng.integers(1, 4))):
            q[rng.integers(0, 4), rng.integers(0, 2)] = bad[int(rng.integers(0, len(bad)))]
        img = base.copy()
        view = img[1:H + 1]
        got = overlay.draw_detected_markers(view, q[None], None, overlay.FIRST_CORNER_LINE8 if trial & 1 else 0)
        assert np.array_equal(img[0], base[0]) and np.array_equal(img[H + 1], base[H + 1])
        if not trial & 1:
            want = odraw.draw_detected_markers(base[1:H + 1].copy(), q[None])
            assert np.array_equal(got, want)
    assert odraw.cv_round(np.nan) == odraw.cv_round(np.inf) == odraw.cv_round(-1e18) == -2**31
    assert odraw.cv_round(2147483520.0) == 2147483520 and odraw.cv_round(-2147483648.0) == -2**31


def test_draw_stride_checks_are_64_bit():
    from fiducials_amd import _lib
    import ctypes as C

    L = _lib.load()
    buf = np.zeros(64, np.uint8)
    # width * 3 wraps to a negative 32-bit number for these widths: the call must be refused, not run
    for w in (715827883, 1431655766, 2**31 - 1):
        rc = L.fid_draw_detected_markers(buf.ctypes.data_as(C.c_void_p), w, 1, 16, None, 0, 0)
        assert rc != 0, w
        rc = L.fid_to_bgr(buf.ctypes.data_as(C.c_void_p), w, 1, 16, 1, buf.ctypes.data_as(C.c_void_p), C.c_int64(64))
        assert rc != 0, w


def test_draw_refuses_bad_arguments():
    img = np.zeros((10, 10, 3), np.uint8)
    with pytest.raises(Exception):
        overlay.draw_detected_markers(img, np.zeros((1, 4, 2), np.float32), None, flags=2)
    with pytest.raises(ValueError):
        overlay.draw_detected_markers(np.zeros((10, 10), np.uint8), np.zeros((1, 4, 2), np.float32))


def test_image_to_bgr8_16bit_and_bayer_encodings():
    """fid_image_to_bgr8 = cv_bridge::toCvCopy(msg, "bgr8") for what a raw camera driver publishes beside the 8-bit colour
    encodings (round-4 review, missing item 5): every 16-bit value through convertTo's rule, both byte orders, padded rows; the
    four Bayer patterns at even and odd sizes against the rules stated independently in oracle/cvbridge.py; and the refusals."""
    from fiducials_amd import _lib
    from oracle import cvbridge as ocb

    rng = np.random.default_rng(21)
    # all 65536 values as one mono16 image: the float product and the tie rule, value by value
    allv = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    got = overlay.image_to_bgr8(allv.view(np.uint8), 256, 256, 512, "mono16")
    assert np.array_equal(got, ocb.to_bgr8_16bit(allv, "mono16"))
    assert got[0, 0, 0] == 0 and got[255, 255, 0] == 255 and got[0, 128, 1] == 0 and got[0, 129, 2] == 1  # 128 * 255 / 65535 = 0.498
    be = allv.byteswap()
    assert np.array_equal(overlay.image_to_bgr8(be.view(np.uint8), 256, 256, 512, "mono16", is_bigendian=True), got)
    for enc, ch in (("bgr16", 3), ("rgb16", 3), ("bgra16", 4), ("rgba16", 4)):
        h, w, pad = 13, 17, 6
        img = rng.integers(0, 65536, (h, w, ch), dtype=np.uint16)
        rows = np.zeros((h, w * ch * 2 + pad), np.uint8)
        rows[:, :w * ch * 2] = img.view(np.uint8).reshape(h, -1)
        assert np.array_equal(overlay.image_to_bgr8(rows, w, h, rows.shape[1], enc), ocb.to_bgr8_16bit(img, enc)), enc
    for enc in ocb.PATTERNS:
        for (h, w) in ((8, 8), (9, 11), (10, 7), (3, 3), (2, 6), (1, 5), (37, 64)):
            raw = rng.integers(0, 256, (h, w + 3), dtype=np.uint8)  # (a padded step)
            got = overlay.image_to_bgr8(raw, w, h, raw.shape[1], enc)
            assert np.array_equal(got, ocb.bayer_to_bgr(raw[:, :w], enc)), (enc, h, w)
        # a flat field of one colour comes back flat in that colour's channel
        flat = np.zeros((12, 12), np.uint8)
        pat = ocb.PATTERNS[enc]
        for y in range(12):
            for x in range(12):
                flat[y, x] = 200 if pat[y & 1][x & 1] == 2 else 0
        out = overlay.image_to_bgr8(flat, 12, 12, 12, enc)
        assert (out[:, :, 2] == 200).all() and (out[:, :, 0] == 0).all() and (out[:, :, 1] == 0).all(), enc
    # the 8-bit encodings are fid_to_bgr
    c3 = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    assert np.array_equal(overlay.image_to_bgr8(c3, 7, 5, 21, "rgb8"), c3[:, :, ::-1])
    # yuv422 (UYVY): every (Y, U, V) on a coarse lattice plus random pixels, a padded step
    ys, us, vs = np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 15), np.arange(0, 256, 15), indexing="ij")
    n = ys.size - (ys.size & 1)
    img = np.zeros((1, n, 2), np.uint8)
    img[0, :, 1] = ys.reshape(-1)[:n]
    img[0, 0::2, 0] = us.reshape(-1)[:n:2]
    img[0, 1::2, 0] = vs.reshape(-1)[:n:2]
    assert np.array_equal(overlay.image_to_bgr8(img, n, 1, 2 * n, "yuv422"), ocb.uyvy_to_bgr(img))
    rnd = rng.integers(0, 256, (9, 14, 2), dtype=np.uint8)
    rows = np.zeros((9, 14 * 2 + 5), np.uint8)
    rows[:, :28] = rnd.reshape(9, 28)
    assert np.array_equal(overlay.image_to_bgr8(rows, 14, 9, rows.shape[1], "yuv422"), ocb.uyvy_to_bgr(rnd))
    gray = np.zeros((1, 4, 2), np.uint8)
    gray[0, :, 0] = 128
    gray[0, :, 1] = (16, 126, 235, 255)
    out = overlay.image_to_bgr8(gray, 4, 1, 8, "yuv422")
    assert out[0, :, 0].tolist() == [0, 128, 255, 255] and (out[0, :, 0] == out[0, :, 2]).all()  # neutral chroma: gray, limited range
    with pytest.raises(FidError):
        overlay.image_to_bgr8(np.zeros((1, 6), np.uint8), 3, 1, 6, "yuv422")  # odd width
    # what is not restated is refused, like the cv_bridge exception the node would catch
    for enc in ("yuv422_yuy2", "bayer_rggb16", "32FC1", ""):
        with pytest.raises(FidError) as e:
            overlay.image_to_bgr8(c3, 7, 5, 21, enc)
        assert e.value.status == _lib.FID_E_UNSUPPORTED
    with pytest.raises(FidError):
        overlay.image_to_bgr8(np.zeros((4, 8), np.uint8), 4, 4, 7, "mono16")  # a step smaller than a row


def test_image_to_bgr8_takes_the_message_bytes_and_refuses_short_buffers():
    """A mono16 frame handed over as a uint16 array is REINTERPRETED (its bytes are the message's data), not value-converted;
    a buffer shorter than step * height never reaches the C side (it reads step * (height - 1) + a row, Bayer two rows ahead)."""
    from fiducials_amd import overlay
    from fiducials_amd._lib import FID_E_INVALID_ARG, FidError

    v = (np.arange(64 * 48, dtype=np.uint32) * 21 % 65536).astype(np.uint16).reshape(48, 64)
    as_bytes = overlay.image_to_bgr8(v.view(np.uint8), 64, 48, 128, "mono16")
    assert np.array_equal(overlay.image_to_bgr8(v, 64, 48, 128, "mono16"), as_bytes)
    assert np.array_equal(overlay.image_to_bgr8(v.tobytes(), 64, 48, 128, "mono16"), as_bytes)
    for enc, data, w, h, step in (("mono16", v.view(np.uint8)[:47], 64, 48, 128), ("bayer_rggb8", np.zeros(12 * 11, np.uint8), 12, 12, 12),
                                  ("bgr8", np.zeros(5 * 21 - 1, np.uint8), 7, 5, 21), ("yuv422", np.zeros(7, np.uint8), 4, 1, 8)):
        with pytest.raises(FidError) as ei:
            overlay.image_to_bgr8(data, w, h, step, enc)
        assert ei.value.status == FID_E_INVALID_ARG, enc


def test_encoding_from_string_is_the_table_the_device_takes():
    """fid_encoding_from_string (ABI 7): sensor_msgs/Image.encoding + is_bigendian -> fid_encoding + bytes per pixel, for every
    string fid_image_to_bgr8 converts; host code, no GPU.  The Python table (_lib.ENC / ENC_BYTES_PER_PIXEL) must say the same."""
    import ctypes as C

    from fiducials_amd import _lib

    L = _lib.load()
    enc, bpp = C.c_int(0), C.c_int32(0)
    for name, value in _lib.ENC.items():
        for be in (0, 1):
            assert L.fid_encoding_from_string(name.encode(), be, C.byref(enc), C.byref(bpp)) == _lib.FID_OK, name
            assert enc.value == _lib.encoding_value(name, bool(be)) and bpp.value == _lib.ENC_BYTES_PER_PIXEL[name], (name, be)
            assert (enc.value & 0x100) == (0x100 if be and name.endswith("16") else 0)
    for bad in (b"bayer_rggb16", b"32FC1", b"", b"MONO8"):
        assert L.fid_encoding_from_string(bad, 0, C.byref(enc), None) == _lib.FID_E_UNSUPPORTED
    assert L.fid_encoding_from_string(None, 0, C.byref(enc), None) == _lib.FID_E_INVALID_ARG
