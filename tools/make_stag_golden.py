#!/usr/bin/env python3
"""tests/golden/stag_hd11_pdf.npz from the reference's only STag fixture: stag_detect/test/test.pdf, 15 pages, each a
1000x1000 JPEG of one HD11 marker with its printed label 00000 ... 00014.  The rasters are stored as the
8-bit gray image the node hands to Stag::detectMarkers (msgToGray: cvtColor RGB2GRAY, stag_ros/utility.hpp:7-17), together
with what the REFERENCE's own detector (oracle/_ref, its sources compiled in place) returns for them with the shipped launch
parameters libraryHD 11 / errorCorrection 2 (stag_detect/launch/stag_detect.launch:9).  Run in the authoring container
(needs /root/reference and Pillow); the output is committed because neither exists on the GPU box."""
import io, os, re, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import stag_ref  # noqa: E402

PDF = "/root/reference/stag_detect/test/test.pdf"
OUT = os.path.join(ROOT, "tests", "golden", "stag_hd11_pdf.npz")


def main():
    d = open(PDF, "rb").read()
    starts = [m.start() for m in re.finditer(b"\xff\xd8\xff", d)]
    assert len(starts) == 15, len(starts)
    grays, ref = [], []
    for page, i in enumerate(starts):
        jpg = d[i:d.find(b"endstream", i)]
        rgb = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))
        assert rgb.shape == (1000, 1000, 3)
        g = oracle.to_gray(rgb, 2)
        grays.append(g)
        m = stag_ref.detect_markers(g, 11, 2)
        assert m.shape[0] == 1 and int(m[0, 0]) == page, (page, m[:, 0])  # the printed label IS the id the reference reads
        ref.append(m[0])
        print("page", page, "-> id", int(m[0, 0]), "corners", np.round(m[0, 1:9], 2))
    np.savez_compressed(OUT, gray=np.stack(grays), labels=np.arange(15, dtype=np.int32), ref_markers=np.stack(ref),
                        library_hd=np.int32(11), error_correction=np.int32(2))
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
