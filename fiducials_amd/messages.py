"""Host-side mirror of the message surface of aruco_detect (§8b): what the node's two
callbacks wrap around the detector -- fiducial_msgs/FiducialArray and FiducialTransformArray (and their ROS 1 wire format,
so that a shim without a ROS installation can still hand over byte-identical messages), the axis-angle -> quaternion step,
and the two string parameters that select fiducials.  Pure host logic: no arithmetic of the hot path lives here.

  FiducialArray            aruco_detect.cpp:342-379      fiducial_msgs/msg/Fiducial.msg, FiducialArray.msg
  FiducialTransformArray   aruco_detect.cpp:397-538      fiducial_msgs/msg/FiducialTransform.msg, FiducialTransformArray.msg
  quaternion               aruco_detect.cpp:447-448,:485 (tf2::Quaternion::setRotation(axis, angle))
  ignore_fiducials         aruco_detect.cpp:540-571      "1,4,8,9-12,30-40"
  fiducial_len_override    aruco_detect.cpp:627-660      "12: 0.2, 100-110: 0.3"
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Header:  # std_msgs/Header
    seq: int = 0
    sec: int = 0
    nsec: int = 0
    frame_id: str = ""


@dataclass
class Fiducial:  # fiducial_msgs/Fiducial: int32 fiducial_id, int32 direction, float64 x0 y0 x1 y1 x2 y2 x3 y3
    fiducial_id: int
    direction: int
    xy: tuple  # (x0, y0, x1, y1, x2, y2, x3, y3)


@dataclass
class FiducialArray:
    header: Header = field(default_factory=Header)
    image_seq: int = 0
    fiducials: list = field(default_factory=list)


@dataclass
class FiducialTransform:  # int32 fiducial_id, geometry_msgs/Transform, float64 image_error object_error fiducial_area
    fiducial_id: int
    translation: tuple
    rotation_xyzw: tuple
    image_error: float
    object_error: float
    fiducial_area: float


@dataclass
class FiducialTransformArray:
    header: Header = field(default_factory=Header)
    image_seq: int = 0
    transforms: list = field(default_factory=list)


def _stoi(s: str) -> int:
    """std::stoi: leading white space, an optional sign, digits; whatever follows is ignored; no digits -> error."""
    t = s.lstrip(" \t\n\r\f\v")
    i = 0
    if i < len(t) and t[i] in "+-":
        i += 1
    j = i
    while j < len(t) and t[j].isdigit():
        j += 1
    if j == i:
        raise ValueError(f"stoi: no conversion in {s!r}")
    return int(t[:j])


def parse_ignore_fiducials(s: str) -> list:
    """FiducialsNode::handleIgnoreString (aruco_detect.cpp:540-571): ids and inclusive ranges, in the order given (duplicates
    kept, as the node's vector does); malformed elements are skipped (the node logs an error)."""
    out = []
    for element in s.split(","):
        if element == "":
            continue
        rng = element.split("-")
        if len(rng) == 2:
            a, b = _stoi(rng[0]), _stoi(rng[1])
            out.extend(range(a, b + 1))
        elif len(rng) == 1:
            out.append(_stoi(rng[0]))
    return out


def parse_fiducial_len_override(s: str) -> dict:
    """The fiducial_len_override parameter (aruco_detect.cpp:627-660): "id: len" or "first-last: len" elements.  As in the node
    the range is split off the WHOLE element (so "100-110: 0.3" reads its upper end from "110: 0.3"), later entries win."""
    out = {}
    for element in s.split(","):
        if element == "":
            continue
        parts = element.split(":")
        if len(parts) != 2:
            continue
        length = float(parts[1])
        rng = element.split("-")
        if len(rng) == 2:
            for j in range(_stoi(rng[0]), _stoi(rng[1]) + 1):
                out[j] = length
        elif len(rng) == 1:
            out[_stoi(rng[0])] = length
    return out


def rvec_to_quaternion(rvec) -> tuple:
    """aruco_detect.cpp:447-448 + tf2::Quaternion::setRotation: angle = |rvec|, axis = rvec / angle,
    q = (axis * sin(angle / 2) / |axis|, cos(angle / 2)); returns (x, y, z, w)."""
    r = np.asarray(rvec, dtype=np.float64)
    angle = math.sqrt(float(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    axis = r / angle
    d = math.sqrt(float(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]))
    s = math.sin(angle * 0.5) / d
    return (float(axis[0] * s), float(axis[1] * s), float(axis[2] * s), math.cos(angle * 0.5))


def make_fiducial_array(image_header: Header, frame_id: str, ids, corners, ignore_ids=()) -> FiducialArray:
    """What imageCallback publishes on fiducial_vertices: stamp and seq of the image, frame of the CameraInfo."""
    fva = FiducialArray(Header(0, image_header.sec, image_header.nsec, frame_id), image_header.seq, [])
    ign = set(ignore_ids)
    for i, c in zip(ids, np.asarray(corners, dtype=np.float64).reshape(-1, 8)):
        if int(i) in ign:
            continue
        fva.fiducials.append(Fiducial(int(i), 0, tuple(float(v) for v in c)))
    return fva


def make_fiducial_transform_array(vertices_header: Header, frame_id: str, ids, poses, ignore_ids=()) -> FiducialTransformArray:
    """What poseEstimateCallback publishes on fiducial_transforms.  poses: the PoseResult of ArucoDetector.pose_last /
    estimate_pose_single_markers (rvecs, tvecs, image_error, object_error, fiducial_area), aligned with ids."""
    fta = FiducialTransformArray(Header(0, vertices_header.sec, vertices_header.nsec, frame_id), vertices_header.seq, [])
    ign = set(ignore_ids)
    for k, i in enumerate(ids):
        if int(i) in ign:
            continue
        q = rvec_to_quaternion(poses.rvecs[k])
        fta.transforms.append(FiducialTransform(int(i), tuple(float(v) for v in poses.tvecs[k]), q, float(poses.image_error[k]),
                                                float(poses.object_error[k]), float(poses.fiducial_area[k])))
    return fta


# ---- ROS 1 wire format (little-endian, packed; arrays = uint32 count + items; strings = uint32 length + bytes)
def _ser_header(h: Header) -> bytes:
    f = h.frame_id.encode()
    return struct.pack("<IIII", h.seq, h.sec, h.nsec, len(f)) + f


def _de_header(b: bytes, p: int):
    seq, sec, nsec, n = struct.unpack_from("<IIII", b, p)
    p += 16
    return Header(seq, sec, nsec, b[p:p + n].decode()), p + n


def serialize_fiducial_array(m: FiducialArray) -> bytes:
    out = [_ser_header(m.header), struct.pack("<iI", m.image_seq, len(m.fiducials))]
    for f in m.fiducials:
        out.append(struct.pack("<ii8d", f.fiducial_id, f.direction, *f.xy))  # 72 bytes
    return b"".join(out)


def deserialize_fiducial_array(b: bytes) -> FiducialArray:
    h, p = _de_header(b, 0)
    image_seq, n = struct.unpack_from("<iI", b, p)
    p += 8
    fs = []
    for _ in range(n):
        v = struct.unpack_from("<ii8d", b, p)
        p += 72
        fs.append(Fiducial(v[0], v[1], tuple(v[2:])))
    assert p == len(b)
    return FiducialArray(h, image_seq, fs)


def serialize_fiducial_transform_array(m: FiducialTransformArray) -> bytes:
    out = [_ser_header(m.header), struct.pack("<iI", m.image_seq, len(m.transforms))]
    for t in m.transforms:
        out.append(struct.pack("<i10d", t.fiducial_id, *t.translation, *t.rotation_xyzw, t.image_error, t.object_error, t.fiducial_area))  # 84 bytes
    return b"".join(out)


def deserialize_fiducial_transform_array(b: bytes) -> FiducialTransformArray:
    h, p = _de_header(b, 0)
    image_seq, n = struct.unpack_from("<iI", b, p)
    p += 8
    ts = []
    for _ in range(n):
        v = struct.unpack_from("<i10d", b, p)
        p += 84
        ts.append(FiducialTransform(v[0], tuple(v[1:4]), tuple(v[4:8]), v[8], v[9], v[10]))
    assert p == len(b)
    return FiducialTransformArray(h, image_seq, ts)
