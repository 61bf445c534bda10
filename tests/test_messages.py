"""Host-side message surface (fiducials_amd/messages.py): byte-identical ROS 1 serialisation checked on the message the
reference node itself recorded (fiducial_slam/test/aruco_transforms.bag, raw bytes kept in tests/golden/golden.json), the
quaternion step, and the two fiducial-selection strings."""
import json
import os

import numpy as np
import pytest

from fiducials_amd import messages as fm

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def test_recorded_transform_array_round_trips_byte_for_byte():
    rec = GOLD["bag_4957"]["transforms"]
    raw = bytes.fromhex(rec["raw_hex"])
    m = fm.deserialize_fiducial_transform_array(raw)
    assert m.image_seq == 4957 and m.header.frame_id == "raspicam" and len(m.transforms) == len(rec["transforms"]) == 7
    assert [t.fiducial_id for t in m.transforms] == [t["fiducial_id"] for t in rec["transforms"]]
    assert len(raw) == 16 + len("raspicam") + 8 + 84 * 7
    assert fm.serialize_fiducial_transform_array(m) == raw
    # built field by field from the parsed values: same bytes again
    m2 = fm.FiducialTransformArray(fm.Header(**rec["header"]), rec["image_seq"],
                                   [fm.FiducialTransform(t["fiducial_id"], tuple(t["translation"]), tuple(t["rotation_xyzw"]), t["image_error"],
                                                         t["object_error"], t["fiducial_area"]) for t in rec["transforms"]])
    assert fm.serialize_fiducial_transform_array(m2) == raw
    for t in m.transforms:  # the recorded rotations are unit quaternions
        assert abs(np.linalg.norm(t.rotation_xyzw) - 1) < 1e-9


def test_fiducial_array_layout_and_ignore_list():
    hdr = fm.Header(4957, 1491682360, 314066469, "whatever")
    corners = np.arange(24, dtype=np.float32).reshape(3, 4, 2) + 0.25
    fva = fm.make_fiducial_array(hdr, "raspicam", [7, 12, 100], corners, ignore_ids=fm.parse_ignore_fiducials("10-12"))
    assert fva.image_seq == 4957 and fva.header.frame_id == "raspicam" and (fva.header.sec, fva.header.nsec) == (hdr.sec, hdr.nsec)
    assert [f.fiducial_id for f in fva.fiducials] == [7, 100] and fva.fiducials[1].xy == tuple(float(v) for v in corners[2].ravel())
    b = fm.serialize_fiducial_array(fva)
    assert len(b) == 16 + 8 + 8 + 72 * 2
    assert fm.deserialize_fiducial_array(b) == fva


def test_quaternion_is_the_rotation_of_the_vector():
    rng = np.random.default_rng(0)
    from fiducials_amd.synth import _rodrigues
    for _ in range(20):
        r = rng.normal(0, 1.2, 3)
        x, y, z, w = fm.rvec_to_quaternion(r)
        assert abs(x * x + y * y + z * z + w * w - 1) < 1e-12
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.abs(R - _rodrigues(r)).max() < 1e-12


def test_selection_strings_parse_like_the_node():
    assert fm.parse_ignore_fiducials("1,4,8,9-12,30-32") == [1, 4, 8, 9, 10, 11, 12, 30, 31, 32]
    assert fm.parse_ignore_fiducials("") == [] and fm.parse_ignore_fiducials(",,5,") == [5]
    assert fm.parse_ignore_fiducials(" 7 , 3-4") == [7, 3, 4]  # stoi skips leading blanks, ignores what follows the digits
    assert fm.parse_ignore_fiducials("1-2-3,9") == [9]  # malformed element: skipped (the node logs an error)
    with pytest.raises(ValueError):
        fm.parse_ignore_fiducials("x")  # std::stoi throws
    assert fm.parse_fiducial_len_override("12: 0.2, 100-102: 0.3") == {12: 0.2, 100: 0.3, 101: 0.3, 102: 0.3}
    assert fm.parse_fiducial_len_override("5: 0.1,5: 0.4") == {5: 0.4}
    assert fm.parse_fiducial_len_override("5 0.1, 6:0.2:3") == {}  # no / two colons: malformed, skipped


def test_transform_array_from_pose_result():
    from fiducials_amd.detector import PoseResult
    n = 3
    poses = PoseResult(rvecs=np.array([[0.1, 0.2, 0.3], [1.0, 0, 0], [0, 0, 2.0]]), tvecs=np.arange(9.0).reshape(3, 3), image_error=np.array([.1, .2, .3]),
                       object_error=np.array([.01, .02, .03]), fiducial_area=np.array([10., 20., 30.]))
    fta = fm.make_fiducial_transform_array(fm.Header(8, 5, 6, "x"), "cam", [4, 5, 6], poses, ignore_ids=[5])
    assert [t.fiducial_id for t in fta.transforms] == [4, 6] and fta.image_seq == 8 and fta.header.frame_id == "cam"
    assert fta.transforms[1].translation == (6.0, 7.0, 8.0) and fta.transforms[1].fiducial_area == 30.0
    assert np.allclose(fta.transforms[1].rotation_xyzw, (0, 0, np.sin(1.0), np.cos(1.0)))
    assert len(fm.serialize_fiducial_transform_array(fta)) == 16 + 3 + 8 + 84 * 2
