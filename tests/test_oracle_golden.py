"""Pin the CPU oracle against every golden vector the reference holds for the aruco path
(SURVEY.md §8c): the reference's own rostest literals and its recorded node output.

These run without a GPU and without /root/reference (fixtures were extracted by tools/make_golden.py).
"""
import json
import os

import numpy as np
import pytest

import oracle
from fiducials_amd.dictionary import get_predefined_dictionary


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "golden.json")))


@pytest.fixture(scope="module")
def d7():
    return get_predefined_dictionary(7)  # node default ~dictionary = 7 (aruco_detect.cpp:611)


def ulp_diff(a, b):
    """distance in float32 ULPs, the metric of gtest's ASSERT_FLOAT_EQ (<= 4 passes)."""
    a = np.asarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def load_gray(golden_dir, key):
    return np.load(os.path.join(golden_dir, key + ".npz"))["gray"]


def test_tag_01(gold, d7, golden_dir):
    """aruco_images_test.cpp:83-110: one marker, id 1, 8 coordinates ASSERT_FLOAT_EQ."""
    ids, corners = oracle.detect(load_gray(golden_dir, "tag_01"), d7)
    assert ids.tolist() == [1]
    ref = np.array(gold["aruco_images_test"]["tag_01"]["1"], dtype=np.float32)
    assert ulp_diff(corners[0].reshape(-1), ref).max() <= 4


def test_tag_245_246(gold, d7, golden_dir):
    """aruco_images_test.cpp:112-153: two markers 245, 246, 16 coordinates ASSERT_FLOAT_EQ."""
    ids, corners = oracle.detect(load_gray(golden_dir, "tag_245_246"), d7)
    assert sorted(ids.tolist()) == [245, 246]
    for i, c in zip(ids, corners):
        ref = np.array(gold["aruco_images_test"]["tag_245_246"][str(int(i))], dtype=np.float32)
        assert ulp_diff(c.reshape(-1), ref).max() <= 4


def _rodrigues(r):
    a = np.linalg.norm(r)
    k = r / a
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx


def test_auto_init_403(gold, d7, golden_dir):
    """auto_init_403_test.cpp:111-138: 403.jpg -> aruco_detect -> fiducial_slam auto-init; the map entry of
    fiducial 403 is T_base_camera * T_camera_fiducial, asserted to 1e-3 in x,y,z and roll,pitch,yaw."""
    g = gold["auto_init_403"]
    cam = gold["aruco_images_test"]
    ids, corners = oracle.detect(load_gray(golden_dir, "img_403"), d7)
    assert ids.tolist() == [403]
    r, t, _ = oracle.solve_pnp_square(cam["K"], cam["D"], corners[0], g["fiducial_len"])
    yaw, pitch, roll = g["base_to_camera_ypr"]

    def Rx(a): return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    def Ry(a): return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    def Rz(a): return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])

    Rbc = Rz(yaw) @ Ry(pitch) @ Rx(roll)
    Rm = Rbc @ _rodrigues(r)
    tm = Rbc @ t + np.array(g["base_to_camera_xyz"])
    p = -np.arcsin(Rm[2, 0])
    rpy = [np.arctan2(Rm[2, 1] / np.cos(p), Rm[2, 2] / np.cos(p)), p, np.arctan2(Rm[1, 0] / np.cos(p), Rm[0, 0] / np.cos(p))]
    assert np.abs(tm - np.array(g["map_xyz"])).max() < 1e-3
    assert np.abs(np.array(rpy) - np.array(g["map_rpy"])).max() < 1e-3


def test_bag_4957_recorded_transforms(gold, d7, golden_dir):
    """aruco_images.bag frame seq 4957 -> the recorded FiducialTransformArray in aruco_transforms.bag
    (real node output): same ids in the same order; translation, quaternion, image_error, object_error
    and fiducial_area per marker.  The recording predates OpenCV 4.2, so the stated tolerance is 1e-6
    relative; in practice 6 of the 7 markers reproduce to < 1e-12."""
    b = gold["bag_4957"]
    gray = load_gray(golden_dir, "bag_4957")
    ids, corners = oracle.detect(gray, d7)
    rec = b["transforms"]["transforms"]
    assert ids.tolist() == [t["fiducial_id"] for t in rec]
    fiducial_len = 0.14  # node default ~fiducial_len (aruco_detect.cpp:612)
    n_exact = 0
    for c, t in zip(corners, rec):
        r, tv, err = oracle.solve_pnp_square(b["K"], b["D"], c, fiducial_len)
        ang = np.linalg.norm(r)
        q = np.concatenate([r / ang * np.sin(ang / 2), [np.cos(ang / 2)]])
        area = oracle.fiducial_area(c)
        obj_err = (err / np.linalg.norm(c[0].astype(np.float64) - c[2])) * (np.linalg.norm(tv) / fiducial_len)
        dq = min(np.abs(q - t["rotation_xyzw"]).max(), np.abs(q + t["rotation_xyzw"]).max())
        dt = np.abs(tv - t["translation"]).max()
        assert dt < 1e-6 and dq < 1e-6
        assert abs(area - t["fiducial_area"]) < 1e-6 * t["fiducial_area"]
        assert abs(err - t["image_error"]) < 1e-4 * max(t["image_error"], 1e-3)
        assert abs(obj_err - t["object_error"]) < 1e-4 * max(t["object_error"], 1e-3)
        n_exact += dt < 1e-12 and dq < 1e-12 and area == t["fiducial_area"]
    assert n_exact >= 6


def test_dictionary_pinned_codewords(d7):
    """dict-7 id 1 bytes {14, 3, 115, 0} and the rotation layout of getByteListFromBits."""
    assert d7.bytes_list[1, 0].tolist() == [14, 3, 115, 0]
    assert d7.pinned[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 245, 246, 403]].all()
    d4 = get_predefined_dictionary(0)
    # DICT_4X4 id 0: all four stored rotations of the published table
    assert d4.bytes_list[0].tolist() == [[181, 50], [235, 72], [76, 173], [18, 215]]
