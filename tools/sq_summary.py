#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* counter_collection.csv files -> per kernel and frame wave-instruction counts (table on stdout,
gpurun_out/pmcsq/sq_summary.json).  Usage: sq_summary.py <frames per launch> <csv> [<csv> ...]"""
import collections
import csv
import json
import sys

B = int(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, path)].add(r["Dispatch_Id"])
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES",
         "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]
out = {"frames_per_launch": B,
       "note": "per kernel: counter sums over all dispatches of the run / (dispatches x frames per launch); one launch = one "
               "sub-batch of frames_per_launch frames; INSTS_* are wave-instructions per frame",
       "kernels": {}}
print("kernel".ljust(30), "disp", *[n[-13:].rjust(14) for n in names])
tot = collections.defaultdict(float)
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    n = max(len(disp[(k, p)]) for p in sys.argv[2:] if (k, p) in disp)
    calls_per_batch = 2 if k == "k_approx" else 1  # two launches of k_approx per sub-batch
    per = {c: acc[k].get(c, 0.0) / (n / calls_per_batch * B) for c in names}
    if k.startswith("k_"):
        for c in names:
            tot[c] += per[c]
    out["kernels"][k] = {"dispatches": n, **{c: round(per[c], 1) for c in names}}
    print(k[:30].ljust(30), str(n).rjust(4), *[f"{per[c]:14.0f}" for c in names])
out["pipeline_per_frame"] = {c: round(tot[c], 1) for c in names}
print("pipeline".ljust(30), "    ", *[f"{tot[c]:14.0f}" for c in names])
json.dump(out, open("gpurun_out/pmcsq/sq_summary.json", "w"), indent=1)
