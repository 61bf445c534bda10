"""The C++ host side (host/: a ROS-free FiducialsNode with the reference's callbacks on top of the C-ABI) run through the
reference's own node test, re-stated without gtest / ROS in host/test/aruco_images_test.cpp: same camera info, same images,
same expected ids and vertices (ASSERT_FLOAT_EQ), plus the recorded bag frame and the node-surface behaviour."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _write_pgm(path, gray):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (gray.shape[1], gray.shape[0]))
        f.write(np.ascontiguousarray(gray, dtype=np.uint8).tobytes())


def _build_host():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    if os.environ.get("FID_HOST_UBSAN") == "1":  # the same tests on the -fsanitize=undefined build (host/Makefile target ubsan)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "ubsan"])
        return os.path.join(ROOT, "host", "bin_ubsan", "aruco_images_test")
    return os.path.join(ROOT, "host", "bin", "aruco_images_test")


def test_host_library_builds_without_a_gpu():
    exe = _build_host()
    assert os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "host", "lib", "libfiducials_host.so"))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout


@pytest.mark.gpu
def test_reference_node_test_passes_on_the_cpp_host(tmp_path):
    exe = _build_host()
    for key in ("tag_01", "tag_245_246", "bag_4957"):
        _write_pgm(tmp_path / f"{key}.pgm", np.load(os.path.join(GOLD, key + ".npz"))["gray"])
    g = json.load(open(os.path.join(GOLD, "golden.json")))["bag_4957"]
    (tmp_path / "bag_4957.txt").write_text(" ".join(repr(float(v)) for v in list(g["K"]) + list(g["D"])[:5]))
    (tmp_path / "bag_4957_msg.hex").write_text(g["transforms"]["raw_hex"])
    try:  # the same frame as it arrives with image_transport `compressed`, and what libjpeg makes of it (the oracle's restatement)
        import io

        from PIL import Image

        from oracle import jpeg as oj
        b = io.BytesIO()
        Image.fromarray(np.load(os.path.join(GOLD, "tag_01.npz"))["gray"]).save(b, "JPEG", quality=90)
        (tmp_path / "tag_01.jpg").write_bytes(b.getvalue())
        bgr = oj.decode(b.getvalue())
        _write_pgm(tmp_path / "tag_01_jpg.pgm", np.ascontiguousarray(bgr[..., 0]))
        # ... and with `format: png` (lossless: the node must publish exactly what the raw frame gives), gray and colour
        g = np.load(os.path.join(GOLD, "tag_01.npz"))["gray"]
        Image.fromarray(g).save(tmp_path / "tag_01_gray.png")
        Image.fromarray(np.stack([g, g, g], axis=-1)).save(tmp_path / "tag_01_rgb.png")
    except ImportError:
        pass  # (no Pillow to write the file: the C++ test skips that check)
    r = subprocess.run([exe, str(tmp_path), os.path.join(ROOT, "fiducials_amd", "data")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_stag_class_and_fiducial_msgs_output_on_the_cpp_host(tmp_path):
    """host/include/stag_host.hpp: class Stag (constructor, detectMarkers, getMarkerList as in stag/Stag.h:41-45), the 5-point
    pose and the fiducial_msgs output, on a rendered HD21 frame."""
    import sys
    sys.path.insert(0, ROOT)
    from fiducials_amd import synth
    from fiducials_amd.stag import load_library
    _build_host()
    fr = synth.make_stag_frame(load_library(21), 8, 1280, 720, 8)
    _write_pgm(tmp_path / "frame.pgm", fr.image)
    lines = [str(len(fr.ids))]
    for i, c, t in zip(fr.ids, fr.corners, fr.tvecs):
        lines.append(" ".join([str(int(i))] + [repr(float(v)) for v in c.ravel()] + [repr(float(t[2]))]))
    (tmp_path / "expected.txt").write_text("\n".join(lines))
    exe = os.path.join(os.path.dirname(_build_host()), "stag_test")
    r = subprocess.run([exe, str(tmp_path / "frame.pgm"), str(tmp_path / "expected.txt"), os.path.join(ROOT, "fiducials_amd", "data"), "21", "7"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_split_call_from_cpp(tmp_path):
    """host/test/batches_in_turn_test.cpp: fid_submit_batch / fid_collect / fid_order_after on two contexts from C++ against
    fid_detect_batch, and the one-batch-per-context rules, on batches made of the reference's tag_01 test image."""
    exe = os.path.join(os.path.dirname(_build_host()), "batches_in_turn_test")
    _write_pgm(tmp_path / "tag_01.pgm", np.load(os.path.join(GOLD, "tag_01.npz"))["gray"])
    r = subprocess.run([exe, str(tmp_path / "tag_01.pgm"), os.path.join(ROOT, "fiducials_amd", "data")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout

