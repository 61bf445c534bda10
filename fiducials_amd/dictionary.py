"""Predefined ArUco dictionaries -- host-side mirror of `aruco::getPredefinedDictionary(dicno)`
(reference: aruco_detect/src/aruco_detect.cpp:671, param `~dictionary` :611).

The enum order is OpenCV's (DICT_4X4_50=0 ... DICT_7X7_1000=15, ARUCO_ORIGINAL=16).  The tables ship
as text files under fiducials_amd/data/ built by tools/make_dictionaries.py: codewords marked P are
authentic OpenCV codewords pinned by the reference's fixtures, codewords marked F are locally generated
fillers (OpenCV's table is third-party data that is neither in the reference nor on this machine --
SURVEY.md §8c).  A deployment that links OpenCV passes `Dictionary::bytesList` straight through the
C-ABI (`fid_dict.bytes`) instead; the byte layout produced here is OpenCV 4.x's
(`Dictionary::getByteListFromBits`: nbytes = ceil(n^2/8), four rotations stored one after another).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# name -> (enum value, marker size, nMarkers, maxCorrectionBits)  [predefined_dictionaries.hpp]
PREDEFINED = {
    "DICT_4X4_50": (0, 4, 50, 1),
    "DICT_4X4_100": (1, 4, 100, 1),
    "DICT_4X4_250": (2, 4, 250, 1),
    "DICT_4X4_1000": (3, 4, 1000, 0),
    "DICT_5X5_50": (4, 5, 50, 3),
    "DICT_5X5_100": (5, 5, 100, 3),
    "DICT_5X5_250": (6, 5, 250, 2),
    "DICT_5X5_1000": (7, 5, 1000, 2),
    "DICT_6X6_50": (8, 6, 50, 6),
    "DICT_6X6_100": (9, 6, 100, 5),
    "DICT_6X6_250": (10, 6, 250, 5),
    "DICT_6X6_1000": (11, 6, 1000, 4),
    "DICT_7X7_50": (12, 7, 50, 9),
    "DICT_7X7_100": (13, 7, 100, 8),
    "DICT_7X7_250": (14, 7, 250, 8),
    "DICT_7X7_1000": (15, 7, 1000, 6),
    "DICT_ARUCO_ORIGINAL": (16, 5, 1024, 0),
}
_BY_ENUM = {v[0]: k for k, v in PREDEFINED.items()}
# OpenCV's N x N tables of 50 / 100 / 250 / 1000 markers are prefixes of one table per size.  The 6 x 6, 7 x 7 and
# ARUCO_ORIGINAL files (and ids >= 250 of 4 x 4) hold labelled fillers only (tools/make_dictionaries.py extra): they exist so
# that every enum value the node accepts (`~dictionary` 0..16) runs, and so that the 5- and 7-byte identify paths are tested.
_FILES = {4: "dict_4x4_1000.txt", 5: "dict_5x5_1000.txt", 6: "dict_6x6_1000.txt", 7: "dict_7x7_1000.txt"}


@dataclass
class Dictionary:
    name: str
    marker_size: int
    max_correction_bits: int
    bytes_list: np.ndarray  # (nMarkers, 4, nbytes) uint8, C-contiguous
    pinned: np.ndarray  # (nMarkers,) bool: codeword is an authentic OpenCV codeword

    @property
    def n_markers(self) -> int:
        return int(self.bytes_list.shape[0])

    def bits(self, marker_id: int) -> np.ndarray:
        """n x n bit matrix (1 = white) of a marker, rotation 0."""
        return bits_from_bytes(self.bytes_list[marker_id, 0], self.marker_size)


def bits_from_bytes(b: np.ndarray, n: int) -> np.ndarray:
    nb = n * n
    out = []
    full = nb // 8
    for i in range(full):
        for k in range(7, -1, -1):
            out.append((int(b[i]) >> k) & 1)
    rem = nb - full * 8
    for k in range(rem - 1, -1, -1):
        out.append((int(b[full]) >> k) & 1)
    return np.array(out, dtype=np.uint8).reshape(n, n)


def byte_list_from_bits(bits: np.ndarray) -> np.ndarray:
    """Dictionary::getByteListFromBits: (4, nbytes) uint8 for rotations 0..3."""
    n = bits.shape[0]
    nbytes = (n * n + 7) // 8
    out = np.zeros((4, nbytes), dtype=np.uint8)
    for r in range(4):
        m = np.rot90(bits, r).reshape(-1)
        cur = 0
        for i, v in enumerate(m):
            out[r, cur] = ((int(out[r, cur]) << 1) | int(v)) & 0xFF
            if i % 8 == 7:
                cur += 1
    return out


def _load_table(n: int, name: str = ""):
    path = os.path.join(_DATA, "dict_aruco_original.txt" if name == "DICT_ARUCO_ORIGINAL" else _FILES[n])
    words, flags = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            i, fl, hx = line.split()
            assert int(i) == len(words)
            words.append(int(hx, 16))
            flags.append(fl == "P")
    return words, flags


_CACHE: dict = {}
# Families whose shipped table holds (next to nothing but) locally generated fillers: under OpenCV's name a detector built on
# them would silently fail to recognise real markers of that family, so they are only handed out on request (arithmetic
# tests: 5- and 7-byte codewords, maxCorrectionBits 0 ... 9).  host/src/fiducials_host.cpp refuses the same enum values.
FILLER_ONLY = frozenset({"DICT_4X4_1000", "DICT_ARUCO_ORIGINAL"} | {k for k in PREDEFINED if k.startswith(("DICT_6X6", "DICT_7X7"))})


def get_predefined_dictionary(which, allow_fillers: bool = False) -> Dictionary:
    """`which` is an OpenCV enum value (the node's `~dictionary` param) or a DICT_* name.  Dictionaries of FILLER_ONLY raise
    unless allow_fillers=True (their codewords are not OpenCV's).  For the 4X4 (<= 250) and 5X5 families `Dictionary.pinned`
    says which ids carry OpenCV's authentic codeword (the ones the reference's fixtures pin); the rest are fillers too --
    a deployment hands over OpenCV's own `Dictionary::bytesList` through `fid_dict.bytes`."""
    name = _BY_ENUM.get(which) if isinstance(which, (int, np.integer)) else which
    if name not in PREDEFINED:
        raise ValueError(f"dictionary {which!r} not available in this build")
    if name in FILLER_ONLY and not allow_fillers:
        raise ValueError(f"dictionary {name} not available in this build: its shipped table holds locally generated filler "
                         "codewords, not OpenCV's (pass allow_fillers=True for arithmetic tests only)")
    if name in _CACHE:
        return _CACHE[name]
    _, n, count, maxc = PREDEFINED[name]
    words, flags = _load_table(n, name)
    bl = np.zeros((count, 4, (n * n + 7) // 8), dtype=np.uint8)
    for i in range(count):
        bits = np.array([(words[i] >> (n * n - 1 - k)) & 1 for k in range(n * n)], dtype=np.uint8).reshape(n, n)
        bl[i] = byte_list_from_bits(bits)
    d = Dictionary(name, n, maxc, np.ascontiguousarray(bl), np.array(flags[:count], dtype=bool))
    _CACHE[name] = d
    return d


def draw_marker(d: Dictionary, marker_id: int, side_pixels: int, border_bits: int = 1) -> np.ndarray:
    """`aruco::drawMarker` semantics (dictionary.cpp Dictionary::drawMarker): (n+2b)^2 cells, black
    border, bit 1 = white, nearest-neighbour upscale to side_pixels (cv::resize INTER_NEAREST)."""
    n = d.marker_size
    cells = n + 2 * border_bits
    tiny = np.zeros((cells, cells), dtype=np.uint8)
    tiny[border_bits:border_bits + n, border_bits:border_bits + n] = d.bits(marker_id) * 255
    # cv::resize INTER_NEAREST: src index = floor(dst * src/dst_size)
    idx = np.minimum((np.arange(side_pixels) * (cells / side_pixels)).astype(np.int64), cells - 1)
    return tiny[np.ix_(idx, idx)]


def load_dictionary_file(path: str, which=-1) -> Dictionary:
    """`aruco::getPredefinedDictionary(which)` from a table file the deployer has (fid_dict_load_file, include/fid_abi.h):
    OpenCV's predefined_dictionaries.hpp as text, a FileStorage YAML (custom dictionaries: which = -1), or a dict_*.txt of this
    repository.  `which`: enum value or DICT_* name.  Every codeword of the result counts as authentic (`pinned`)."""
    import ctypes as C

    from . import _lib

    dicno = which if isinstance(which, (int, np.integer)) else PREDEFINED[which][0]
    L = _lib.load()
    fd = _lib.FidDict()
    rc = L.fid_dict_load_file(path.encode(), int(dicno), None, 0, C.byref(fd))
    if rc != _lib.FID_E_CAPACITY:
        raise _lib.FidError(rc, (L.fid_dict_last_error() or b"").decode())
    nbytes = (fd.marker_size * fd.marker_size + 7) // 8
    buf = np.zeros((fd.n_markers, 4, nbytes), dtype=np.uint8)
    rc = L.fid_dict_load_file(path.encode(), int(dicno), buf.ctypes.data, buf.nbytes, C.byref(fd))
    if rc != _lib.FID_OK:
        raise _lib.FidError(rc, (L.fid_dict_last_error() or b"").decode())
    name = _BY_ENUM.get(int(dicno), f"custom_{fd.marker_size}x{fd.marker_size}_{fd.n_markers}")
    return Dictionary(name, fd.marker_size, fd.max_correction_bits, buf, np.ones(fd.n_markers, dtype=bool))
