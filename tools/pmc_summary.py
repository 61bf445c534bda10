#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/gpu_pmc3.sh into profiles/pmc_traffic.json: calibration factors from the
known-byte-count kernels (tools/pmc_calib.hip), then FETCH_SIZE / WRITE_SIZE per kernel and frame with the factor of the
kernel's dominant access pattern applied.  Usage: pmc_summary.py <outdir> <frames> <libfid_amd.so> > pmc_traffic.json"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (fiducials_amd._lib.device_text_sha256)

out, frames, lib = sys.argv[1], int(sys.argv[2]), sys.argv[3]
KNOWN = 1 << 30


def read(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


cal_f, cal_w = read("calib_FETCH_SIZE"), read("calib_WRITE_SIZE")
# FETCH_SIZE / WRITE_SIZE are reported in KiB
factors = {}
for k, v in cal_f.items():
    if k.startswith("rd"):
        factors[k] = KNOWN / (1024.0 * (sum(v) / len(v)))
for k, v in cal_w.items():
    if k.startswith("wr"):
        factors[k] = KNOWN / (1024.0 * (sum(v) / len(v)))
# which calibrated pattern stands for a kernel's reads / writes (its dominant access)
READ_AS = {"k_threshold_stream": "rd1", "k_threshold_fixed": "rd4", "k_find_starts": "rd16", "k_walk_full": "rdlds", "k_seed_walk": "rdlds",
           "k_probe": "rd4"}
WRITE_AS = {"k_walk_full": "wr16", "k_seed_walk": "wr16", "k_seg_copy": "wr4"}
pf, pw = read("FETCH_SIZE"), read("WRITE_SIZE")
kernels = {}
for k in sorted(set(pf) | set(pw)):
    if "rocclr" in k:
        continue
    base = k.split("<")[0]
    rf = factors.get(READ_AS.get(base, "rd4"), 1.0)
    wf = factors.get(WRITE_AS.get(base, "wr4"), 1.0)
    # the bench run makes 2 calls (warm-up + 1 step): mean over the dispatches of the kernel, times launches per call
    nd = max(len(pf.get(k, [])), len(pw.get(k, [])), 1)
    per_call = nd / 2.0
    fb = sum(pf.get(k, [0])) / max(len(pf.get(k, [1])), 1) * 1024 * per_call
    wb = sum(pw.get(k, [0])) / max(len(pw.get(k, [1])), 1) * 1024 * per_call
    kernels[k] = {"fetch_bytes_per_frame_raw": round(fb / frames), "write_bytes_per_frame_raw": round(wb / frames),
                  "fetch_bytes_per_frame": round(fb * rf / frames), "write_bytes_per_frame": round(wb * wf / frames),
                  "read_pattern": READ_AS.get(base, "rd4"), "write_pattern": WRITE_AS.get(base, "wr4")}


def stage(*names):
    return sum(v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"] for k, v in kernels.items() if any(k.startswith(n) for n in names))


doc = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (one counter per run, kernel trace only), "
              "bench.py --batch %d, one sub-batch; calibration kernels tools/pmc_calib.hip (1 GiB each, known byte counts); "
              "script tools/gpu_pmc3.sh" % frames,
    "library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
    "device_text_sha256": __import__("fiducials_amd._lib", fromlist=["_lib"]).device_text_sha256(lib),
    "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; bytes = value * 1024 * calibration factor",
    "calibration": {"known_bytes": KNOWN, "factor_true_over_counter": {k: round(v, 4) for k, v in sorted(factors.items())},
                    "note": "factor = bytes the calibration kernel really moved / bytes the counter reports, per access pattern; a "
                            "pipeline kernel is corrected with the factor of its dominant pattern (read_pattern / write_pattern)"},
    "per_frame_bytes": {
        "threshold": stage("k_threshold"), "find_starts": stage("k_find_starts"), "walk_probe": stage("k_probe"),
        "walk_full": stage("k_walk_full<2>", "k_seg_"), "seed_walk": stage("k_walk_full<1>", "k_seed_walk"), "approx": stage("k_approx"),
    },
    "frames_per_launch_measured": frames,  # the passes ran on ONE sub-batch of this many frames; bench.py scales bytes per frame
                                           # onto its own launches (128 frames at cfg 3)
    "pipeline_bytes_per_frame": sum(v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"] for v in kernels.values()),
    "algorithmic_bytes_per_frame": {"threshold": 5443200, "masks": 3369600, "pipeline": 8812800},
    "kernels": kernels,
}
print(json.dumps(doc, indent=1))
