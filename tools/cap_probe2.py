import os, sys, faulthandler
sys.path.insert(0, '/root/repo')
os.environ["FID_TRACE"] = sys.argv[1]
faulthandler.dump_traceback_later(25, exit=True)
import numpy as np
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.synth import make_frame
from fiducials_amd._lib import FidError
d = get_predefined_dictionary(6)
fr = make_frame(d, 3, width=1280, height=720, n_markers=8)
for kw in (dict(max_contours=96), dict(max_points=4096), dict()):
    det = ArucoDetector(6, max_width=1280, max_height=720, **kw)
    try:
        c, i = det.detect_markers(fr.image); print(kw, "ok", len(i), flush=True)
    except FidError as e:
        print(kw, "status", e.status, flush=True)
    print("  counts", det.tap_counts()[0].tolist(), flush=True)
    det.close()
