import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from fiducials_amd.detector import ArucoDetector, default_params
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.synth import make_frame
from test_gpu_parity import PARAM_SETS
k, dic = 7, 6
d = get_predefined_dictionary(dic)
fr = make_frame(d, 300 + 7 * k + dic, width=1280, height=720, n_markers=10, side_range=(70, 130))
p, op = default_params(), oracle.default_params()
for name, v in PARAM_SETS[k].items():
    setattr(p, name, v); setattr(op, name, v)
det = ArucoDetector(d, params=p, max_width=1280, max_height=720)
corners, ids = det.detect_markers(fr.image)
oids, ocorners = oracle.detect(fr.image, d, params=op)
pre = det.tap_presubpix()[0][:len(ids)]["corners"].reshape(-1, 4, 2)
for i in range(len(ids)):
    dd = np.abs(corners[i] - ocorners[i]).max()
    print(i, ids[i], oids[i], dd, "pre", pre[i].tolist() if dd > 0 else "")
    if dd > 0:
        print(" gpu", corners[i].tolist()); print(" ora", ocorners[i].tolist())
