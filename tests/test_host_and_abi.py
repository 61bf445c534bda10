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
    declared = set(re.findall(r"\b(fid_[a-z_]+)\s*\(", hdr))
    L = _lib.load()
    for s in declared:
        assert hasattr(L, s), s
    assert declared == set(_lib.SYMBOLS)
    assert L.fid_abi_version() == 5
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
