import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from fiducials_amd import stag as fstag, synth
words=fstag.load_library(21)
fr=synth.make_stag_frame(words,100,1920,1080,20).image
det=fstag.StagDetector(21,7,max_width=1920,max_height=1080)
for i in range(2):
    M=det.detect_markers(fr)
print(len(M))
if len(sys.argv) > 1:
    segs = det.edge_segments(validated=True)
    ln = sorted((len(p) for p in segs), reverse=True)
    print("validated segments", len(ln), "longest", ln[:12], "over 1024:", sum(l > 1024 for l in ln), "over 256:", sum(l > 256 for l in ln))
    L = det.lines(validated=False)
    print("lines", len(L))
