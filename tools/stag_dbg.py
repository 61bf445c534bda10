import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fiducials_amd import stag as fstag
from oracle import stag_ref
from test_gpu_stag import QUAD_CASES, _quads_as_table
np.set_printoptions(precision=6, linewidth=200, suppress=True)
for case in sys.argv[1:]:
    img = QUAD_CASES[case]()
    det = fstag.StagDetector(21, 7, max_width=1920, max_height=1080)
    det.detect_quads(img)
    ref, ng = stag_ref.detect_quads(img)
    got = _quads_as_table(det.quads())
    print(case, got.shape, ref.shape, "corner groups", ng)
    for i in range(min(len(got), len(ref))):
        if not (got[i] == ref[i]).all():
            print(" quad", i, "\n  got", got[i], "\n  ref", ref[i], "\n  diff", got[i] - ref[i])
    det.close()
