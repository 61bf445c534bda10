"""Host-side logic and the C-ABI surface, without a GPU: the library loads, exports every symbol
include/fid_abi.h declares, refuses to run without a device (no CPU fallback), and the dictionary /
synthetic-frame helpers behave deterministically."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from fiducials_amd import _lib
from fiducials_amd.dictionary import byte_list_from_bits, draw_marker, get_predefined_dictionary
from fiducials_amd.synth import make_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fid_abi.h")).read()
    declared = set(re.findall(r"\b(fid_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.load()
    for s in declared:
        assert hasattr(L, s), s
    assert declared == set(_lib.SYMBOLS)
    assert L.fid_abi_version() == 7  # (== FID_ABI_VERSION of include/fid_abi.h: __graft_entry__.build() compares the two)
    assert b"no CPU fallback" in L.fid_strerror(_lib.FID_E_NO_DEVICE)


def test_struct_layouts_match_header():
    # sizes the C side was compiled with (fid_abi.h): natural alignment, no packing
    assert C.sizeof(_lib.FidMarker) == 36
    assert C.sizeof(_lib.FidPoseOut) == 72
    assert C.sizeof(_lib.FidCandidate) == 52
    assert C.sizeof(_lib.FidLimits) == 32
    assert C.sizeof(_lib.FidParams) == 128
    p = _lib.FidParams()
    _lib.load().fid_default_params(C.byref(p))
    # node defaults (aruco_detect.cpp:690-727), not OpenCV's
    assert (p.adaptiveThreshWinSizeMin, p.adaptiveThreshWinSizeMax, p.adaptiveThreshWinSizeStep) == (3, 53, 4)
    assert p.cornerRefinementMethod == 1 and p.cornerRefinementMinAccuracy == 0.01
    assert p.minMarkerPerimeterRate == 0.1 and p.polygonalApproxAccuracyRate == 0.01
    assert p.perspectiveRemovePixelPerCell == 8 and p.maxErroneousBitsInBorderRate == 0.04


def test_no_gpu_means_loud_failure_not_fallback():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fiducials_amd.detector import ArucoDetector
    with pytest.raises(_lib.FidError) as e:
        ArucoDetector(7)
    assert e.value.status == _lib.FID_E_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "fiducials_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in src and "liboracle" not in src and "aruco_oracle.h" not in src, f


def test_dictionary_layout_and_draw_marker():
    d = get_predefined_dictionary("DICT_5X5_250")
    assert d.bytes_list.shape == (250, 4, 4) and d.max_correction_bits == 2
    bits = d.bits(1)
    assert "/".join("".join(map(str, r)) for r in bits) == "00001/11000/00001/10111/00110"
    bl = byte_list_from_bits(bits)
    assert np.array_equal(bl, d.bytes_list[1])
    assert np.array_equal(byte_list_from_bits(np.rot90(bits, 1))[0], bl[1])
    img = draw_marker(d, 1, 70)
    assert img.shape == (70, 70) and img[:10].max() == 0 and img[15, 55] == 255  # border black, bit (0,4) white
    # 5X5_250 is a prefix of 5X5_1000 (OpenCV stores one table)
    d1000 = get_predefined_dictionary(7)
    assert np.array_equal(d1000.bytes_list[:250], d.bytes_list)


def test_synth_is_deterministic_and_annotated():
    d = get_predefined_dictionary(6)
    a = make_frame(d, 123, width=640, height=480, n_markers=4, side_range=(60, 110))
    b = make_frame(d, 123, width=640, height=480, n_markers=4, side_range=(60, 110))
    assert np.array_equal(a.image, b.image) and np.array_equal(a.corners, b.corners)
    assert a.image.dtype == np.uint8 and a.corners.shape == (4, 4, 2)
    c = make_frame(d, 124, width=640, height=480, n_markers=4, side_range=(60, 110))
    assert not np.array_equal(a.image, c.image)


# ---- fid_dict_load_file: a dictionary table the deployer has (OpenCV's header as text, a FileStorage YAML, a dict_*.txt)
def _as_opencv_header(d5, d4, aruco=None):
    """Text in the layout of OpenCV's modules/aruco/src/predefined_dictionaries.hpp, filled from tables of this repository."""
    def arr(name, bl):
        nb = bl.shape[2]
        rows = ["    { " + ", ".join("{ " + ", ".join(str(int(b)) for b in rot) + " }" for rot in m) + ", }," for m in bl]
        return f"static unsigned char {name}[][4][{nb}] = {{\n" + "\n".join(rows) + "\n};\n"
    txt = "/* a comment with braces { } and numbers 1 2 3 */\n// DICT_4X4_1000_BYTES mentioned in a comment first\n"
    txt += arr("DICT_4X4_1000_BYTES", d4) + arr("DICT_5X5_1000_BYTES", d5)
    if aruco is not None:
        txt += arr("DICT_ARUCO_BYTES", aruco)
    return txt


def test_dict_load_file_opencv_header_yaml_and_txt(tmp_path):
    from fiducials_amd.dictionary import load_dictionary_file
    d5 = get_predefined_dictionary("DICT_5X5_1000")
    d4 = get_predefined_dictionary("DICT_4X4_1000", allow_fillers=True)
    hpp = tmp_path / "predefined_dictionaries.hpp"
    hpp.write_text(_as_opencv_header(d5.bytes_list, d4.bytes_list))
    for name in ("DICT_5X5_50", "DICT_5X5_250", "DICT_5X5_1000", "DICT_4X4_100", "DICT_4X4_1000"):
        got = load_dictionary_file(str(hpp), name)
        want = get_predefined_dictionary(name, allow_fillers=True)
        assert (got.marker_size, got.max_correction_bits, got.n_markers) == (want.marker_size, want.max_correction_bits, want.n_markers)
        assert np.array_equal(got.bytes_list, want.bytes_list)
    # the first two rows of DICT_4X4_1000_BYTES as OpenCV's header prints them (the rotation / bit packing convention)
    first = load_dictionary_file(str(hpp), 0).bytes_list
    assert first[0].tolist() == [[181, 50], [235, 72], [76, 173], [18, 215]]
    assert first[1].tolist() == [[15, 154], [101, 71], [89, 240], [226, 166]]
    with pytest.raises(_lib.FidError):
        load_dictionary_file(str(hpp), "DICT_6X6_50")  # that array is not in the file
    with pytest.raises(_lib.FidError):
        load_dictionary_file(str(hpp), -1)  # a header holds many tables
    # FileStorage YAML (Dictionary::writeDictionary): a custom 6 x 6 dictionary of 7 markers
    rng = np.random.default_rng(4)
    bits = rng.integers(0, 2, (7, 6, 6))
    y = tmp_path / "custom.yml"
    y.write_text("%YAML:1.0\n---\nnmarkers: 7\nmarkersize: 6\nmaxCorrectionBits: 4\n" +
                 "".join(f'marker_{i}: "{"".join(str(int(b)) for b in bits[i].reshape(-1))}"\n' for i in range(7)))
    c = load_dictionary_file(str(y), -1)
    assert (c.marker_size, c.max_correction_bits, c.n_markers) == (6, 4, 7)
    for i in range(7):
        assert np.array_equal(c.bytes_list[i], byte_list_from_bits(bits[i].astype(np.uint8)))
        assert np.array_equal(c.bits(i), bits[i])
    # this repository's own text tables
    t = load_dictionary_file(os.path.join(ROOT, "fiducials_amd", "data", "dict_5x5_1000.txt"), 6)
    assert np.array_equal(t.bytes_list, get_predefined_dictionary(6).bytes_list)
    with pytest.raises(_lib.FidError):
        load_dictionary_file(str(tmp_path / "missing.hpp"), 0)


def test_catkin_node_parses_against_the_ros_stubs():
    """ros/aruco_detect_amd/src/aruco_detect_amd_node.cpp through g++ -fsyntax-only -Wall -Wextra -Werror against stand-in ROS
    headers (fiducial_msgs and DetectorParamsConfig generated from the reference's own .msg / .cfg) and the real host/ headers."""
    import subprocess
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "ros"), "syntax"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
