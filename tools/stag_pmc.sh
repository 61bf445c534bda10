#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only) of the STag pipeline in group mode (frames as a grid
# dimension): the cfg 5 bench frames, 64 per call, two calls -> gpurun_out/pmc_stag/stag_pmc_traffic.json (copy to profiles/).
# FETCH_SIZE under-reports by the factor calibrated in profiles/pmc_traffic.json (2.0 on this part), WRITE_SIZE is exact.
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmc_stag
rm -rf $OUT; mkdir -p $OUT
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  STAG_CHILD=1 CTX=64 B=64 STEPS=1 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python tools/gpu_stag_batch.py > $OUT/$c.log 2>&1
done
python - <<'PY'
import csv, collections, glob, hashlib, json, re
FRAMES = 128  # two calls of 64
doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one counter per run), fid_stag_detect_markers_batch in group mode, "
                 "64 frame slots, 2 calls x 64 frames of the cfg 5 bench frames (tools/stag_pmc.sh)",
       "library_sha256": hashlib.sha256(open("fiducials_amd/lib/libfid_amd.so", "rb").read()).hexdigest(),
       "device_text_sha256": __import__("fiducials_amd._lib", fromlist=["_lib"]).device_text_sha256("fiducials_amd/lib/libfid_amd.so"),
       "frames_measured": FRAMES, "fetch_factor": 2.0, "write_factor": 1.0,
       "units": "counters in KiB; bytes = value * 1024 * factor (FETCH_SIZE reports half of the bytes on this part: profiles/pmc_traffic.json calibration)",
       "kernels": {}}
tot = 0.0
for c, fac in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    acc = collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmc_stag/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"k_stag_batch<(k_stag_\w+)_fn>", k)
            acc[m.group(1) if m else k.split("(")[0].replace("void ", "")[:40]] += float(r["Counter_Value"]) * 1024 * fac
    for k, v in acc.items():
        doc["kernels"].setdefault(k, {})[c.lower() + "_bytes_per_frame"] = round(v / FRAMES)
        tot += v / FRAMES
doc["pipeline_bytes_per_frame"] = round(tot)
doc["algorithmic_bytes_per_frame"] = 10 * 1920 * 1080
json.dump(doc, open("gpurun_out/pmc_stag/stag_pmc_traffic.json", "w"), indent=1)
print("pipeline bytes per frame", round(tot), "=", round(tot / (10 * 1920 * 1080), 2), "x algorithmic")
for k, v in sorted(doc["kernels"].items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    print(k.ljust(30), v)
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
