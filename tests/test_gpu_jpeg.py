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


_KEEP = {}


def keeping_decoder(w, h):
    """A decoder whose coefficients stay readable after a call (FID_JPEG_KEEP_COEFS at creation): by default the IDCT zeroes what
    it has read, so that the next call finds the array clean without a fill -- the decoder under test works that way, and the
    fixtures going through it one after the other check that the array IS clean every time."""
    if (w, h) not in _KEEP:
        os.environ["FID_JPEG_KEEP_COEFS"] = "1"
        try:
            _KEEP[(w, h)] = fj.JpegDecoder(max_width=w, max_height=h)
        finally:
            del os.environ["FID_JPEG_KEEP_COEFS"]
    return _KEEP[(w, h)]


def check(dec, data):
    bgr_o, coefs_o, planes_o = oj.decode(data, stages=True)
    keep = keeping_decoder(dec.max_width, dec.max_height)
    assert np.array_equal(keep.decode(data, "bgr8"), bgr_o), "bgr (coefficients kept)"
    assert np.array_equal(keep.tap(fj.TAP_COEFS), coefs_o), "coefficients"
    got = dec.decode(data, "bgr8")
    with pytest.raises(FidError):
        dec.tap(fj.TAP_COEFS)  # consumed by the IDCT
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


def _encode(img, **kw):
    PIL = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    PIL.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def test_full_size_frames_batches_and_the_detector():
    """1920 x 1080 marker frames as compressed_image_transport sends them (libjpeg defaults: 4:2:0, quality 80) and in the other
    layouts, in one batch: every frame `==` the oracle's decode (and libjpeg-turbo's, where Pillow is there to ask); the
    detector fed with the device-resident gray output finds what the oracle detector finds in the oracle's decode."""
    import oracle
    from fiducials_amd.detector import ArucoDetector
    from fiducials_amd.dictionary import get_predefined_dictionary
    from fiducials_amd.synth import make_frame

    PIL = pytest.importorskip("PIL.Image")
    d = get_predefined_dictionary(6)
    frames = [make_frame(d, 300 + k).image for k in range(4)]
    rgb = [np.stack([f, np.roll(f, 3, 1), 255 - f // 2], -1) for f in frames]
    files = [
        _encode(rgb[0], quality=80, subsampling=2),                              # compressed_image_transport's default
        _encode(rgb[1], quality=95, subsampling=1),
        _encode(rgb[2], quality=60, subsampling=0, restart_marker_blocks=7),
        _encode(frames[3], quality=90),                                          # one component
        _encode(rgb[3], quality=80, subsampling=2, restart_marker_rows=1),
    ]
    dec = fj.JpegDecoder(max_width=1920, max_height=1080, max_batch=len(files))
    det = ArucoDetector(d, max_width=1920, max_height=1080, max_batch=len(files), max_markers=64)
    try:
        got = dec.decode(files, "bgr8")
        rounds = dec.last_rounds()
        assert 1 <= rounds <= 12, rounds  # sub-sequences fall into step within a few rounds
        for k, f in enumerate(files):
            want = oj.decode(f)
            assert np.array_equal(got[k], want), k
            ref = np.asarray(PIL.open(io.BytesIO(f)).convert("RGB"))[..., ::-1]
            assert np.array_equal(got[k], ref), k
        # gray, left on the device, straight into the detector
        dec.decode(files, "mono8", to_host=False)
        ptr, w, h, stride, fstride = dec.device_ptr()
        assert (w, h, stride, fstride) == (1920, 1080, 1920, 1920 * 1080)
        res = det.detect_markers_device(ptr, len(files), w, h)
        for k, f in enumerate(files):
            oids, ocorners = oracle.detect(gray_of(oj.decode(f)), d)
            assert res[k][1].tolist() == oids.tolist(), k
            assert np.array_equal(res[k][0], ocorners), k
        assert sum(len(r[1]) for r in res) >= 60  # (the markers survive the compression)
    finally:
        dec.close()
        det.close()


def test_odd_sizes_and_damaged_files():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    dec = fj.JpegDecoder(max_width=700, max_height=520)
    try:
        for (w, h) in [(641, 479), (333, 17), (9, 301), (700, 520)]:
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            img[h // 3:h // 2] //= 6
            for sub in (0, 1, 2):
                for rst in (0, 5):
                    kw = dict(quality=int(rng.integers(20, 99)), subsampling=sub)
                    if rst:
                        kw["restart_marker_blocks"] = rst
                    check(dec, _encode(img, **kw))
        # the corners of the format: quality 1 (long zero runs, ZRL) and 100 (long codes: the 16-bit tables), tables optimised
        # for the image (not the standard ones), a restart marker behind every MCU
        img = rng.integers(0, 256, (203, 311, 3)).astype(np.uint8)
        img[50:120, 60:200] = (img[50:120, 60:200] // 16) + 200
        for kw in (dict(quality=1, subsampling=2), dict(quality=100, subsampling=0), dict(quality=100, subsampling=2, optimize=True),
                   dict(quality=35, subsampling=1, optimize=True), dict(quality=75, subsampling=2, restart_marker_blocks=1),
                   dict(quality=90, subsampling=0, restart_marker_blocks=1), dict(quality=50, subsampling=2, restart_marker_rows=2)):
            check(dec, _encode(img, **kw))
        check(dec, _encode(img[..., 0], quality=97, optimize=True))
        data = _encode(rng.integers(0, 256, (240, 320, 3)).astype(np.uint8), quality=85)
        cut = data[:len(data) // 2]
        a = dec.decode(cut, "bgr8")
        assert a.shape == (240, 320, 3) and np.array_equal(a, dec.decode(cut, "bgr8"))
        bad = bytearray(data)
        for p in range(len(bad) // 2, len(bad) // 2 + 40):
            bad[p] = 0x5A
        b = dec.decode(bytes(bad), "bgr8")
        assert b.shape == (240, 320, 3) and np.array_equal(b, dec.decode(bytes(bad), "bgr8"))
        with pytest.raises(FidError):
            dec.decode(_encode(np.zeros((600, 800, 3), np.uint8)))  # larger than the context
    finally:
        dec.close()


def test_randomised_sweep_equals_libjpeg_turbo():
    """tools/gpu_jpeg_stress.py in small (400 files: 0 mismatches when it was run in full): random sizes, contents, qualities,
    layouts, restart intervals, standard and optimised tables, mixed in batches; the device decode `==` libjpeg-turbo's."""
    import subprocess
    import sys

    pytest.importorskip("PIL.Image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_jpeg_stress.py"), "60", "77"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout


def test_damaged_entropy_data_ends_in_an_image_or_a_status():
    """A frame damaged on its way (compressed_image_transport over a lossy link): bits flipped, bytes overwritten, data cut
    or doubled BEHIND the headers.  The decoder starts every sub-sequence out of step by design, so arbitrary bits are its
    normal diet: a call returns an image of the announced size or a status -- it neither hangs nor writes outside its buffers
    (the next, sound, file on the same context must still decode bit for bit)."""
    gold = np.load(GOLD)
    rng = np.random.default_rng(21)
    dec = fj.JpegDecoder(max_width=256, max_height=256)
    try:
        cases = [k for k in gold["cases"].tolist() if len(gold[f"jpg_{k[0]}"]) - gold[f"jpg_{k[0]}"].tobytes().rfind(b"\xff\xda") > 300]
        assert len(cases) >= 6
        ok = bad = 0
        for it in range(160):
            k, w, h = cases[it % len(cases)][:3]
            data = bytearray(gold[f"jpg_{k}"].tobytes())
            sos = bytes(data).rfind(b"\xff\xda")
            assert sos > 0
            lo = sos + 14  # behind the scan header
            kind = it % 4
            if kind == 0:
                for _ in range(int(rng.integers(1, 12))):
                    data[int(rng.integers(lo, len(data) - 2))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                at = int(rng.integers(lo, len(data) - 2))
                n = int(rng.integers(1, 48))
                data[at:at + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            elif kind == 2:
                del data[int(rng.integers(lo, len(data) - 2)):-2]  # cut, EOI kept
            else:
                at = int(rng.integers(lo, len(data) - 2))
                data[at:at] = data[at:at + int(rng.integers(1, 200))]
            try:
                img = dec.decode(bytes(data), "bgr8")
                assert img.shape == (h, w, 3)
                ok += 1
            except FidError as e:
                assert e.status in (1, 3, 4, 6)  # INVALID_ARG / HIP (did not settle) / CAPACITY / UNSUPPORTED
                bad += 1
            if it % 16 == 15:  # a sound file in between: bit for bit
                kk = cases[(it // 16) % len(cases)][0]
                assert np.array_equal(dec.decode(gold[f"jpg_{kk}"].tobytes(), "bgr8"), gold[f"bgr_{kk}"])
        assert ok + bad == 160 and ok > 40
    finally:
        dec.close()
