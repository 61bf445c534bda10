import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fiducials_amd import stag as fstag
from test_gpu_stag import ROUTE_CASES
case = sys.argv[1]
img = ROUTE_CASES[case]()
res = {}
for mode in ("seq", "par"):
    os.environ["FID_STAG_ROUTE"] = mode
    det = fstag.StagDetector(21, 7, max_width=1920, max_height=1080)
    det.detect_edges(img)
    res[mode] = (det.tap(fstag.TAP_EDGEIMG).copy(), det.tap(fstag.TAP_SEGMENTS).reshape(-1, 2).copy(), det.tap(fstag.TAP_SEGPIX).reshape(-1, 2).copy())
    det.close()
es, ss, ps = res["seq"]; ep, sp, pp = res["par"]
print("edge equal", np.array_equal(es, ep), "nseg", len(ss), len(sp), "npix", len(ps), len(pp))
for i in range(min(len(ss), len(sp))):
    a = ps[ss[i,0]:ss[i,0]+ss[i,1]]; b = pp[sp[i,0]:sp[i,0]+sp[i,1]]
    if not (np.array_equal(ss[i], sp[i]) and np.array_equal(a, b)):
        print("first diff at segment", i, "seq (first,len)", ss[i], "par", sp[i])
        print(" prev seg seq", ss[i-1] if i else None, "last pix", ps[ss[i-1,0]+ss[i-1,1]-1] if i else None, " memory before block seq", ps[ss[i,0]-2:ss[i,0]+1].tolist())
        print(" seq head", a[:6].tolist(), "\n par head", b[:6].tolist())
        print(" seq tail", a[-4:].tolist(), "\n par tail", b[-4:].tolist())
        break
