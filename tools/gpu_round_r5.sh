#!/bin/bash
# Round 5 evidence.  Part A (counters keyed to the library hash; their files go into profiles/ BEFORE part B, because bench.py
# reads them): PMC traffic (aruco + STag), SQ instruction counters incl. the lane-utilisation pass.  Part B: full GPU test suite,
# host UBSan run, stress sweep, the default bench line, the --feed host and --feed jpeg lines, kernel stats (aruco, STag, JPEG) with
# a sidecar that names the library they were measured on, smoke.  Part C: the short form of B.   Usage: gpu_round_r5.sh A | B | C
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r5final; mkdir -p $OUT
SHA=$(sha256sum fiducials_amd/lib/libfid_amd.so | cut -d' ' -f1)
if [ "${1:-A}" = "A" ]; then
  bash tools/gpu_pmc3.sh > $OUT/pmc3.log 2>&1; tail -4 $OUT/pmc3.log | cut -c1-200; cp gpurun_out/pmc3/pmc_traffic.json $OUT/pmc_traffic.json
  bash tools/gpu_pmc_sq.sh > $OUT/sq.log 2>&1; tail -3 $OUT/sq.log | cut -c1-200; cp gpurun_out/pmcsq/sq_summary.json $OUT/sq_cycles.json
  bash tools/stag_pmc.sh > $OUT/stag_pmc.log 2>&1; tail -2 $OUT/stag_pmc.log | cut -c1-200; cp gpurun_out/pmc_stag/stag_pmc_traffic.json $OUT/stag_pmc_traffic.json 2>/dev/null
  echo $SHA | tee $OUT/lib.sha256
  exit 0
fi
if [ "${1:-A}" = "C" ]; then
  # the short form after a change to the STag kernels only: full GPU suite, default bench line, aruco + STag kernel stats, smoke
  ( time timeout 900 python -m pytest tests -m gpu -q --timeout 400 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
  ( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-300 $OUT/bench.json
  rm -rf $OUT/prof; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
  python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.csv; head -6 $OUT/kernel_stats.csv | cut -c1-110; rm -rf $OUT/prof
  STAG_CHILD=1 CTX=64 B=128 STEPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/profs -o r -- python tools/gpu_stag_batch.py > $OUT/profs.log 2>&1
  python tools/rocpd_stats.py $(find $OUT/profs -name '*.db' | head -1) > $OUT/stag_kernel_stats.csv; head -6 $OUT/stag_kernel_stats.csv | cut -c1-110; rm -rf $OUT/profs
  bash tools/gpu_trace_stag.sh > $OUT/stag_single_trace.log 2>&1; tail -3 $OUT/stag_single_trace.log
  ( timeout 120 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
  NO_REF=1 timeout 200 python tools/stag_bench.py > $OUT/stag_single.log 2>&1; tail -1 $OUT/stag_single.log
  python - <<PY
import json, sys
sys.path.insert(0, "/root/repo")
from fiducials_amd import _lib
json.dump({"library_sha256": "$SHA", "device_text_sha256": _lib.device_text_sha256(), "files": ["r05_kernel_stats.csv", "r05_stag_kernel_stats.csv", "r05_stag_single_trace.log"],
           "commands": {"r05_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras",
                        "r05_stag_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python tools/gpu_stag_batch.py (CTX=64 B=128 STEPS=3)",
                        "r05_stag_single_trace.log": "tools/gpu_trace_stag.sh (one cfg 5 frame queued ahead, kernel by kernel)"},
           "note": "r05_jpeg_kernel_stats.csv, r05_host_ubsan.log, r05_stress*.log, r05_bench_feed_*.json were measured on the library before the round's STag kernel work (device text 441b68415b71...): the aruco, JPEG and host code they exercise is unchanged since"},
          open("$OUT/kernel_stats.json", "w"), indent=1)
PY
  echo $SHA | tee $OUT/lib.sha256
  exit 0
fi
( time timeout 900 python -m pytest tests -m gpu -q --timeout 400 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
( FID_HOST_UBSAN=1 timeout 300 python -m pytest tests/test_gpu_host_cpp.py -q -m gpu ) > $OUT/host_ubsan.log 2>&1; echo "rc=$?" >> $OUT/host_ubsan.log; tail -3 $OUT/host_ubsan.log
( timeout 600 python tools/gpu_stress.py 200 ) > $OUT/stress.log 2>&1; tail -2 $OUT/stress.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-300 $OUT/bench.json
( timeout 300 python bench.py --feed host --steps 12 --warmup 3 --no-extras --no-cpu-baseline ) 2> /dev/null | grep '^{' > $OUT/bench_feed_host.json; cut -c1-200 $OUT/bench_feed_host.json
( timeout 300 python bench.py --feed jpeg --steps 12 --warmup 3 --no-extras --no-cpu-baseline ) 2> /dev/null | grep '^{' > $OUT/bench_feed_jpeg.json; cut -c1-200 $OUT/bench_feed_jpeg.json
rm -rf $OUT/prof; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-110
rm -rf $OUT/prof
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
STAG_CHILD=1 CTX=64 B=128 STEPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/profs -o r -- python tools/gpu_stag_batch.py > $OUT/profs.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profs -name '*.db' | head -1) > $OUT/stag_kernel_stats.csv; head -8 $OUT/stag_kernel_stats.csv | cut -c1-110; tail -1 $OUT/profs.log | cut -c1-150
rm -rf $OUT/profs
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profj -o r -- python tools/gpu_jpeg_bench.py 256 80 > $OUT/profj.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profj -name '*.db' | head -1) > $OUT/jpeg_kernel_stats.csv; head -6 $OUT/jpeg_kernel_stats.csv | cut -c1-110
rm -rf $OUT/profj
( timeout 120 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
NO_REF=1 timeout 200 python tools/stag_bench.py > $OUT/stag_single.log 2>&1; tail -1 $OUT/stag_single.log
python - <<PY
import json, hashlib
sha = "$SHA"
import sys; sys.path.insert(0, "/root/repo")
from fiducials_amd import _lib
json.dump({"library_sha256": sha, "device_text_sha256": _lib.device_text_sha256(), "files": ["r05_kernel_stats.csv", "r05_stag_kernel_stats.csv", "r05_jpeg_kernel_stats.csv"],
           "commands": {"r05_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras",
                        "r05_stag_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python tools/gpu_stag_batch.py (CTX=64 B=128 STEPS=3)",
                        "r05_jpeg_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python tools/gpu_jpeg_bench.py 256 80"}},
          open("$OUT/kernel_stats.json", "w"), indent=1)
PY
echo $SHA | tee $OUT/lib.sha256
