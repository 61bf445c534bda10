import os, sys
sys.path.insert(0, '/root/repo')
mode, kind = sys.argv[1], sys.argv[2]
os.environ["FID_TRACE"] = mode
import numpy as np
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.synth import make_frame
from fiducials_amd._lib import FidError
d = get_predefined_dictionary(6)
fr = make_frame(d, 3, width=1280, height=720, n_markers=8)
kw = dict(max_contours=96) if kind == "contours" else dict(max_points=4096)
det = ArucoDetector(6, max_width=1280, max_height=720, **kw)
try:
    det.detect_markers(fr.image); print(mode, kind, "NO ERROR")
except FidError as e:
    print(mode, kind, "status", e.status, e)
