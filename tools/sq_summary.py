#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* counter_collection.csv files -> per kernel and frame wave-instruction counts (table on stdout,
gpurun_out/pmcsq/sq_summary.json).  Usage: sq_summary.py <frames per launch> <csv> [<csv> ...]
A kernel that is launched several times per call (k_approx: two launches on each of two streams; k_seg_cycles / k_seg_copy: one
per stream) is summed over its launches: the divisor is the number of CALLS = the dispatches of k_sort_cands (once per call)."""
import collections
import csv
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (fiducials_amd._lib.device_text_sha256)

B = int(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
lane = collections.defaultdict(lambda: [0.0, 0.0])  # kernel -> [thread cycles, active-instruction cycles] of the pass that has both
seen = collections.defaultdict(set)  # counter -> passes that carry it (a counter collected in two passes is averaged, not added)
for path in sys.argv[2:]:
    rows = list(csv.DictReader(open(path)))
    both = {"SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"} <= {r["Counter_Name"] for r in rows}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[r["Counter_Name"]].add(path)
        disp[(k, path)].add(r["Dispatch_Id"])
        if both and r["Counter_Name"] == "SQ_THREAD_CYCLES_VALU":
            lane[k][0] += float(r["Counter_Value"])
        if both and r["Counter_Name"] == "SQ_ACTIVE_INST_VALU":
            lane[k][1] += float(r["Counter_Value"])
for k in acc:
    for c in acc[k]:
        acc[k][c] /= max(len(seen[c]), 1)
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES",
         "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE",
         "SQ_THREAD_CYCLES_VALU"]
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fiducials_amd", "lib", "libfid_amd.so")
out = {"frames_per_launch": B,
       "library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,
       "device_text_sha256": __import__("fiducials_amd._lib", fromlist=["_lib"]).device_text_sha256(lib) if os.path.exists(lib) else None,
       "trace_mode": os.environ.get("FID_TRACE", "cycles"),
       "note": "per kernel: counter sums over all dispatches of the run / (calls x frames per call); a call = one sub-batch of "
               "frames_per_launch frames through the whole pipeline; kernels launched several times per call are summed over "
               "their launches (launches_per_call); INSTS_* are wave-instructions per frame",
       "kernels": {}}
calls = max((len(v) for (k, p), v in disp.items() if k == "k_sort_cands"), default=1)
out["calls"] = calls
print("kernel".ljust(30), "disp", *[n[-13:].rjust(14) for n in names])
tot = collections.defaultdict(float)
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    n = max(len(disp[(k, p)]) for p in sys.argv[2:] if (k, p) in disp)
    per = {c: acc[k].get(c, 0.0) / (calls * B) for c in names}
    if k.startswith("k_"):
        for c in names:
            tot[c] += per[c]
    out["kernels"][k] = {"dispatches": n, "launches_per_call": round(n / calls, 2), **{c: round(per[c], 1) for c in names}}
    lu = lane[k][0] / (64.0 * lane[k][1]) if lane[k][1] > 0 else None
    if lu is not None:
        # lanes that execute per VALU wave-instruction / 64 (rocprofiler's VALUUtilization / 100), both counters from ONE pass
        out["kernels"][k]["lane_util"] = round(lu, 4)
        if k.startswith("k_"):
            tot["_lane_num"] += lane[k][0]
            tot["_lane_den"] += 64.0 * lane[k][1]
    print(k[:30].ljust(30), str(n).rjust(4), *[f"{per[c]:14.0f}" for c in names], "" if lu is None else f" lanes {lu:.3f}")
out["pipeline_per_frame"] = {c: round(tot[c], 1) for c in names}
if tot["_lane_den"] > 0:
    out["pipeline_lane_util"] = round(tot["_lane_num"] / tot["_lane_den"], 4)  # weighted by every kernel's VALU instructions
print("pipeline".ljust(30), "    ", *[f"{tot[c]:14.0f}" for c in names])
json.dump(out, open("gpurun_out/pmcsq/sq_summary.json", "w"), indent=1)
