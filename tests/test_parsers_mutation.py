"""The library's host-side parsers of UNTRUSTED input -- frames as compressed_image_transport hands them over (JPEG headers:
fid_jpeg_probe; PNG files: fid_png_probe / fid_png_decode) and a deployer's dictionary table file (fid_dict_load_file) -- under
random damage: every call either succeeds or returns an error status (FidError); none may crash, hang or hand back an image
of a size other than the one it announced.  (The reference node wraps cv::imdecode / cv_bridge in try / catch and drops the
frame, aruco_detect.cpp:389-394; a C-ABI cannot throw, so the statuses are the contract.)  No GPU needed: all three are host code.
"""
import io
import os

import numpy as np
import pytest

from fiducials_amd import _lib
from fiducials_amd import jpeg as fj
from fiducials_amd import png as fpng
from fiducials_amd.dictionary import get_predefined_dictionary, load_dictionary_file

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def _damage(data: bytes, rng) -> bytes:
    """One of: flipped bits, a run of random bytes, a cut, a doubled slice, a 16-bit field blown up."""
    b = bytearray(data)
    kind = int(rng.integers(0, 5))
    if kind == 0:
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        at = int(rng.integers(0, len(b)))
        n = int(rng.integers(1, 32))
        b[at:at + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 2:
        b = b[:int(rng.integers(0, len(b)))]
    elif kind == 3:
        at = int(rng.integers(0, len(b)))
        b[at:at] = b[at:at + int(rng.integers(1, 64))]
    else:
        at = int(rng.integers(0, max(1, len(b) - 2)))
        b[at:at + 2] = b"\xff\xff"
    return bytes(b)


def test_jpeg_probe_survives_damaged_headers():
    gold = np.load(GOLD)
    files = [gold[k].tobytes() for k in gold.files if k.startswith("jpg_")]
    assert len(files) >= 8
    rng = np.random.default_rng(11)
    ok = bad = 0
    for it in range(1500):
        data = _damage(files[it % len(files)], rng)
        try:
            info = fj.probe(data)
            assert 0 < info["width"] <= 65535 and 0 < info["height"] <= 65535
            ok += 1
        except _lib.FidError as e:
            assert e.status in (_lib.FID_E_INVALID_ARG, 6)  # INVALID_ARG or UNSUPPORTED, never anything else
            bad += 1
    assert ok > 50 and bad > 50  # (the sweep reaches both sides)
    with pytest.raises(_lib.FidError):
        fj.probe(b"")
    with pytest.raises(_lib.FidError):
        fj.probe(b"\xff\xd8")


def _damage_png_behind_the_crc(data: bytes, rng) -> bytes:
    """Damage INSIDE a chunk with its CRC recomputed: what the container check lets through to the header fields, the
    palette, inflate and the row filters."""
    import struct
    import zlib

    chunks, at = [], 8
    while at + 12 <= len(data):
        n = struct.unpack(">I", data[at:at + 4])[0]
        chunks.append((data[at + 4:at + 8], bytearray(data[at + 8:at + 8 + n])))
        at += 12 + n
    k = int(rng.integers(0, len(chunks)))
    typ, body = chunks[k]
    if len(body):
        for _ in range(int(rng.integers(1, 4))):
            body[int(rng.integers(0, len(body)))] = int(rng.integers(0, 256))
        if rng.random() < 0.2:
            del body[int(rng.integers(0, len(body))):]
    out = bytearray(data[:8])
    for t, b in chunks:
        out += struct.pack(">I", len(b)) + t + bytes(b) + struct.pack(">I", zlib.crc32(t + bytes(b)) & 0xffffffff)
    return bytes(out)


def test_png_decode_survives_damaged_files():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(12)
    files = []
    for mode, shape in (("L", (37, 53)), ("RGB", (24, 31, 3)), ("RGBA", (16, 16, 4)), ("P", (20, 20))):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        im = PIL.fromarray(a if mode != "P" else a, "L" if mode == "P" else mode)
        if mode == "P":
            im = im.convert("P")
        buf = io.BytesIO()
        im.save(buf, "PNG")
        files.append(buf.getvalue())
    ok = bad = 0
    for it in range(1600):
        src = files[it % len(files)]
        data = _damage(src, rng) if it % 2 else _damage_png_behind_the_crc(src, rng)
        try:
            info = fpng.probe(data)
        except _lib.FidError:
            bad += 1
            continue
        if info["width"] * info["height"] > (1 << 22):  # (a blown-up IHDR: the decoder has its own cap; keep the test's memory small)
            with pytest.raises(_lib.FidError):
                fpng.decode(data, "mono8")
            bad += 1
            continue
        try:
            img = fpng.decode(data, "mono8")
            assert img.shape == (info["height"], info["width"])
            ok += 1
        except _lib.FidError:
            bad += 1
    assert ok > 20 and bad > 200  # (every chunk carries a CRC: plain damage is caught there, the re-signed kind goes deeper)


def test_dict_load_file_survives_damaged_tables(tmp_path):
    d = get_predefined_dictionary("DICT_5X5_250")
    lines = ["static unsigned char DICT_5X5_1000_BYTES[][4][4] = {"]
    full = get_predefined_dictionary("DICT_5X5_1000")
    for m in range(full.n_markers):
        lines.append("    { " + ", ".join("{ " + ", ".join(str(int(v)) for v in full.bytes_list[m, r]) + " }" for r in range(4)) + " },")
    lines.append("};")
    hpp = "\n".join(lines).encode()
    yml = ("%YAML:1.0\n---\nnmarkers: 3\nmarkersize: 4\nmaxCorrectionBits: 1\n" +
           "".join(f'marker_{i}: "{"".join(str((i * 7 + k) % 2) for k in range(16))}"\n' for i in range(3))).encode()
    rng = np.random.default_rng(13)
    ok = bad = 0
    for it in range(400):
        src, which = ((hpp, 6), (yml, -1))[it % 2]
        p = tmp_path / f"t{it % 4}.txt"
        p.write_bytes(_damage(src, rng))
        try:
            got = load_dictionary_file(str(p), which)
            assert got.bytes_list.shape[0] == got.n_markers and got.bytes_list.shape[1] == 4
            if which == 6:
                assert (got.marker_size, got.n_markers) == (d.marker_size, d.n_markers)
            ok += 1
        except _lib.FidError:
            bad += 1
    assert ok > 20 and bad > 20


def test_dict_yaml_absurd_count_is_refused_quickly_and_order_does_not_matter(tmp_path):
    """Round-4 advisor: 'nmarkers: 2000000000' zero-filled tens of GB before the first marker_<i> look-up failed, and every
    marker was searched from the top of the file.  The count is now bounded by what the file can hold, the search resumes
    behind the previous marker, and a file whose markers are out of order still loads."""
    import time

    marks = [f'marker_{i}: "{"".join(str((i * 5 + k) % 2) for k in range(16))}"\n' for i in range(6)]
    p = tmp_path / "absurd.yml"
    p.write_bytes(("%YAML:1.0\n---\nnmarkers: 2000000000\nmarkersize: 4\nmaxCorrectionBits: 1\n" + "".join(marks)).encode())
    t = time.perf_counter()
    with pytest.raises(_lib.FidError):
        load_dictionary_file(str(p), -1)
    assert time.perf_counter() - t < 1.0
    fwd = tmp_path / "fwd.yml"
    fwd.write_bytes(("%YAML:1.0\n---\nnmarkers: 6\nmarkersize: 4\nmaxCorrectionBits: 1\n" + "".join(marks)).encode())
    rev = tmp_path / "rev.yml"
    rev.write_bytes(("%YAML:1.0\n---\nnmarkers: 6\nmarkersize: 4\nmaxCorrectionBits: 1\n" + "".join(reversed(marks))).encode())
    a, b = load_dictionary_file(str(fwd), -1), load_dictionary_file(str(rev), -1)
    assert a.n_markers == b.n_markers == 6 and np.array_equal(a.bytes_list, b.bytes_list)
    big = tmp_path / "big.txt"
    with open(big, "wb") as fh:  # larger than the 64 MB cap: refused, not silently cut off
        fh.write(b"# filler\n" * (8 << 20))
    with pytest.raises(_lib.FidError):
        load_dictionary_file(str(big), 6)
