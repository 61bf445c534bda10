#!/usr/bin/env python3
"""Extracts the STag marker libraries (the codeword tables the STag authors published with the detector) from the reference's
data header into fiducials_amd/data/stag_HD<hd>.bin (raw little-endian uint64).

Provenance: /root/reference/stag_detect/include/stag/MarkerIDs.h -- arrays HD11 ... HD23 of 48-bit codewords, four
pre-rotated copies of every marker (`Decoder::Decoder`, stag_detect/src/stag/Decoder.cpp:14-43: noOfCodewords = len / 4;
id = i % noOfCodewords, shift = i / noOfCodewords).  These are DATA (like OpenCV's predefined ArUco dictionaries): the
detector is useless without the published tables.  Only the numbers are taken.  Run where /root/reference is mounted; the .bin files are committed."""
import os
import re
import sys

import numpy as np

SRC = "/root/reference/stag_detect/include/stag/MarkerIDs.h"
DST_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fiducials_amd", "data")
EXPECT = {11: 22309, 13: 2884, 15: 766, 17: 157, 19: 38, 21: 12, 23: 6}  # Decoder.cpp:17-37


def main():
    text = open(SRC).read()
    out = {}
    for m in re.finditer(r"HD(\d+)\[(\d+)\]\s*=\s*\{([^}]*)\}", text):
        hd, n = int(m.group(1)), int(m.group(2))
        vals = np.array([int(v) for v in re.findall(r"\d+", m.group(3))], dtype=np.uint64)
        assert len(vals) == n == 4 * EXPECT[hd], (hd, len(vals), n)
        assert (vals < (1 << 48)).all()
        out[f"HD{hd}"] = vals
    assert sorted(out) == sorted(f"HD{k}" for k in EXPECT)
    for k, v in out.items():  # raw little-endian uint64, read by fiducials_amd/stag.py and host/include/stag_host.hpp alike
        v.astype("<u8").tofile(os.path.join(DST_DIR, f"stag_{k}.bin"))
    print("wrote", DST_DIR, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
