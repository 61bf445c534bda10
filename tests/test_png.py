"""fid_png_decode (host code of the library: frames that arrive as PNG through compressed_image_transport) against libpng --
Pillow's decoder -- on files Pillow writes (every colour type / bit depth it can save, adaptive row filters), on files built
here with ONE forced row filter each, on the reference's own test images (the golden gray images under tests/golden were made
from them through Pillow), and on damaged / unsupported files.  PNG is lossless: a correct decoder returns the same bytes.
No GPU: these entry points are host-only."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

from fiducials_amd import png as fpng
from fiducials_amd._lib import FID_E_INVALID_ARG, FID_E_UNSUPPORTED, FidError

Image = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gray_of_bgr(bgr):  # cvtColor(BGR2GRAY), the constants of k_to_gray / oracle.to_gray
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def _pil_bgr(data):
    """What cv::imdecode(IMREAD_COLOR) returns, through libpng: RGB of the file (alpha dropped, palette expanded, gray
    replicated, 16 bit -> high byte), channels swapped."""
    im = Image.open(io.BytesIO(data))
    if im.mode in ("I;16", "I;16B", "I"):
        a = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)  # png_set_strip_16
        rgb = np.stack([a, a, a], axis=-1)
    else:
        rgb = np.asarray(im.convert("RGBA").convert("RGB") if im.mode in ("LA", "PA", "RGBA", "P") and "A" in im.mode else im.convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


def _check(data):
    want = _pil_bgr(data)
    got = fpng.decode(data, "bgr8")
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(fpng.decode(data, "mono8"), _gray_of_bgr(want))


def _texture(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (xx * 3 + yy * 2) % 256
    img = np.stack([(base * (k + 1) + rng.integers(0, 24, (h, w))) % 256 for k in range(c)], axis=-1).astype(np.uint8)
    img[h // 4:h // 2, w // 4:w // 2] = 255 - img[h // 4:h // 2, w // 4:w // 2]  # a flat-ish patch: other filters win there
    return img[..., 0] if c == 1 else img


@pytest.mark.parametrize("mode,c", [("L", 1), ("RGB", 3), ("RGBA", 4), ("LA", 2)])
@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 48), (333, 127)])
def test_files_pillow_writes(mode, c, size):
    w, h = size
    arr = _texture(h, w, c, 7 * w + c)
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, "PNG")
    _check(buf.getvalue())
    info = fpng.probe(buf.getvalue())
    assert (info["width"], info["height"], info["bit_depth"], info["interlace"]) == (w, h, 8, 0)
    assert info["gray"] == (1 if mode in ("L", "LA") else 0)


def test_palette_low_bit_depths_and_sixteen_bits():
    rgb = _texture(40, 61, 3, 3)
    for colors in (2, 4, 16, 200):  # 1, 2, 4, 8 bit palette indices
        buf = io.BytesIO()
        Image.fromarray(rgb).quantize(colors).save(buf, "PNG")
        assert fpng.probe(buf.getvalue())["color_type"] == 3
        _check(buf.getvalue())
    g = _texture(33, 70, 1, 5)
    buf = io.BytesIO()
    Image.fromarray((g > 127).astype(np.uint8) * 255).convert("1").save(buf, "PNG")  # 1-bit gray
    assert fpng.probe(buf.getvalue())["bit_depth"] == 1
    _check(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray((g.astype(np.uint16) * 257 + 13)).save(buf, "PNG")  # 16-bit gray: the high byte survives
    assert fpng.probe(buf.getvalue())["bit_depth"] == 16
    _check(buf.getvalue())


def _chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)


def _encode(arr, bit_depth, filter_type):
    """A PNG writer for the test: every row with the same filter type (the PNG specification's section 9, forward direction)."""
    h = arr.shape[0]
    rows = arr.reshape(h, -1).astype(np.uint8)
    bpp = max(1, (arr.shape[2] if arr.ndim == 3 else 1) * bit_depth // 8)
    out = bytearray()
    prev = np.zeros(rows.shape[1], dtype=np.int64)
    for y in range(h):
        cur = rows[y].astype(np.int64)
        left = np.concatenate([np.zeros(bpp, np.int64), cur[:-bpp]]) if cur.size > bpp else np.zeros_like(cur)
        upleft = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]]) if cur.size > bpp else np.zeros_like(cur)
        if filter_type == 0:
            f = cur
        elif filter_type == 1:
            f = cur - left
        elif filter_type == 2:
            f = cur - prev
        elif filter_type == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        out.append(filter_type)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    return zlib.compress(bytes(out), 6)


@pytest.mark.parametrize("ft", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("c,ct", [(1, 0), (3, 2), (4, 6)])
def test_every_row_filter_on_its_own(ft, c, ct):
    w, h = 53, 29
    arr = _texture(h, w, c, 11 * ft + c)
    z = _encode(arr.reshape(h, w, c), 8, ft)
    pieces = [z[:len(z) // 3], z[len(z) // 3:len(z) // 3 + 1], z[len(z) // 3 + 1:]]  # the zlib stream cut across three IDAT chunks
    data = (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ct, 0, 0, 0)) + _chunk(b"tEXt", b"Comment\0x")
            + b"".join(_chunk(b"IDAT", p) for p in pieces) + _chunk(b"IEND", b""))
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), np.asarray(Image.fromarray(arr).convert("RGB")))  # the writer is right
    _check(data)


def test_the_reference_test_images_decode_to_the_golden_gray_images():
    """aruco_detect/test/test_images/*.png (what aruco_images_test.cpp feeds the node): fid_png_decode(MONO8) == the gray image
    tools/make_golden.py stored (Pillow's decode + the oracle's BGR2GRAY).  The files themselves are not in this repository."""
    ref = "/root/reference/aruco_detect/test/test_images/"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not here")
    for key, name in (("tag_01", "tag_01_d7_14cm.png"), ("tag_245_246", "tag_245-246_d7_14cm.png")):
        data = open(ref + name, "rb").read()
        gold = np.load(os.path.join(GOLD, key + ".npz"))
        assert np.array_equal(fpng.decode(data, "mono8"), gold["gray"])
        x0, y0, x1, y1 = gold["crop_xyxy"]
        assert np.array_equal(fpng.decode(data, "bgr8")[y0:y1, x0:x1, ::-1], gold["rgb_crop"])


def test_damaged_and_unsupported_files_are_refused():
    buf = io.BytesIO()
    Image.fromarray(_texture(20, 30, 3, 1)).save(buf, "PNG")
    good = buf.getvalue()
    cases = {
        "signature": b"\x89PNX" + good[4:],
        "crc": good[:40] + bytes([good[40] ^ 1]) + good[41:],
        "truncated": good[:len(good) // 2],
        "no IEND": good[:-12],
    }
    for name, data in cases.items():
        with pytest.raises(FidError) as e:
            fpng.decode(data)
        assert e.value.status == FID_E_INVALID_ARG, name
    # a stream that inflates to fewer rows than the header says
    z = zlib.compress(b"\0" + bytes(30 * 3))
    short = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 30, 20, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", z) + _chunk(b"IEND", b"")
    with pytest.raises(FidError):
        fpng.decode(short)
    inter = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 30, 20, 8, 2, 0, 0, 1)) + _chunk(b"IDAT", z) + _chunk(b"IEND", b"")
    with pytest.raises(FidError) as e:
        fpng.decode(inter)
    assert e.value.status == FID_E_UNSUPPORTED
    assert fpng.probe(inter)["interlace"] == 1  # (the header parse itself accepts it)
    with pytest.raises(ValueError):
        fpng.decode(good, "rgb8")
