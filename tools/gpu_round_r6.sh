#!/bin/bash
# Round 6 evidence, in parts (each a gpurun call; the counter files of part A go into profiles/ BEFORE part B, because bench.py
# reads them and refuses a file measured on another build):
#   O = the occupancy sheet: kernel traces (STag batch, STag group alone, aruco bench, aruco single frame) + the library's own launch
#       log (dynamic LDS) + one SQ pass each -> occupancy.json, stag_occupancy.json, aruco_single_trace.log
#   A = counters keyed to the library hash: PMC traffic (aruco + STag), SQ instruction counters incl. lane utilisation
#   B = full GPU test suite, host UBSan run, stress sweep, the default / --feed host / --feed jpeg bench lines, kernel stats (aruco,
#       STag, JPEG), STag single-frame trace, queue-ahead stress, smoke
# Usage: gpu_round_r6.sh O | A | B      -> gpurun_out/r6final/
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r6final; mkdir -p $OUT
SHA=$(sha256sum fiducials_amd/lib/libfid_amd.so | cut -d' ' -f1)
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag')); bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
if [ "${1:-O}" = "O" ]; then
  W=$OUT/occ; rm -rf $W; mkdir -p $W
  export FID_LAUNCH_LOG=$PWD/$W/launch_log.txt
  PMC="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES"
  STAG_CHILD=1 NOQ=1 CTX=256 B=1024 STEPS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $W/st -o r -- python tools/gpu_stag_batch.py > $W/stag_trace.log 2>&1
  cp $(find $W/st -name '*kernel_trace.csv' | head -1) $W/stag_batch_trace.csv; rm -rf $W/st; grep fps $W/stag_trace.log | tail -1
  STAG_CHILD=1 NOQ=1 CTX=256 B=512 STEPS=1 timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $W/stc -o r -- python tools/gpu_stag_batch.py > $W/stag_pmc.log 2>&1
  mkdir -p $W/stag_group_mode_counters; cp $(find $W/stc -name '*counter_collection.csv' | head -1) $W/stag_group_mode_counters/counters.csv 2>/dev/null; rm -rf $W/stc
  STAG_CHILD=1 NOQ=1 CTX=32 B=96 STEPS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $W/st1 -o r -- python tools/gpu_stag_batch.py > $W/stag_trace1.log 2>&1
  cp $(find $W/st1 -name '*kernel_trace.csv' | head -1) $W/stag_group_alone_trace.csv; rm -rf $W/st1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $W/ar -o r -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $W/aruco_trace.log 2>&1
  cp $(find $W/ar -name '*kernel_trace.csv' | head -1) $W/aruco_bench_trace.csv; rm -rf $W/ar
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $W/arc -o r -- python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $W/aruco_pmc.log 2>&1
  mkdir -p $W/aruco_counters; cp $(find $W/arc -name '*counter_collection.csv' | head -1) $W/aruco_counters/counters.csv 2>/dev/null; rm -rf $W/arc
  unset FID_LAUNCH_LOG
  bash tools/gpu_trace1.sh > $OUT/aruco_single_trace.log 2>&1; tail -30 $OUT/aruco_single_trace.log
  python tools/occupancy.py --label "round 6 final library" --launch-log $W/launch_log.txt --trace $W/stag_batch_trace.csv $W/aruco_bench_trace.csv $W/stag_group_alone_trace.csv \
     --check $W/stag_group_mode_counters/counters.csv $W/aruco_counters/counters.csv > $OUT/occupancy.json 2> $W/occupancy.err; tail -2 $W/occupancy.err
  python - <<PY
import json
d = json.load(open("$OUT/occupancy.json"))
keep = ("k_stag_route_walk[g]", "k_stag_route_extract[g]", "k_stag_route_extract_small[g]", "k_stag_refine[g]", "k_stag_split_lines[g]", "k_stag_quads[g]",
        "k_stag_validate_lines[g]", "k_stag_ccl_flatten[g]", "k_stag_ccl_tile[g]", "k_stag_smooth_grad[g]", "k_stag_smooth3_prewitt[g]", "k_stag_place[g]", "k_stag_anchors[g]")
out = {"library_sha256": d["library_sha256"], "device_text_sha256": d["device_text_sha256"],
       "what": "the latency-bound STag kernels in GROUP mode (cfg 5: 256 slots = 8 groups of 32): launch shape and residency by resource from the code "
               "object + the library's launch log, and beside it ONE measured SQ pass (rocprofv3 serialises the dispatches of a counter run: a kernel ALONE on the chip)",
       "sheet": {k: d["kernels"][k] for k in keep if k in d["kernels"]},
       "measured_alone": {k: d.get("measured", {}).get("stag_group_mode_counters", {}).get(k) for k in keep},
       "coresidency": {k: v for k, v in d.get("coresidency", {}).items() if "stag" in k}}  # (per trace file)
json.dump(out, open("$OUT/stag_occupancy.json", "w"), indent=1)
PY
  cp $W/launch_log.txt $OUT/launch_log.txt; gzip -9 -f $W/*_trace.csv; rm -f $W/*/counters.csv; ls -la $OUT $W | head -40
  echo $SHA | tee $OUT/lib.sha256
  exit 0
fi
if [ "${1:-O}" = "A" ]; then
  bash tools/gpu_pmc3.sh > $OUT/pmc3.log 2>&1; tail -4 $OUT/pmc3.log | cut -c1-200; cp gpurun_out/pmc3/pmc_traffic.json $OUT/pmc_traffic.json
  bash tools/gpu_pmc_sq.sh > $OUT/sq.log 2>&1; tail -3 $OUT/sq.log | cut -c1-200; cp gpurun_out/pmcsq/sq_summary.json $OUT/sq_cycles.json
  bash tools/stag_pmc.sh > $OUT/stag_pmc.log 2>&1; tail -2 $OUT/stag_pmc.log | cut -c1-200; cp gpurun_out/pmc_stag/stag_pmc_traffic.json $OUT/stag_pmc_traffic.json 2>/dev/null
  echo $SHA | tee $OUT/lib.sha256
  exit 0
fi
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 600 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
( FID_HOST_UBSAN=1 timeout 300 python -m pytest tests/test_gpu_host_cpp.py -q -m gpu ) > $OUT/host_ubsan.log 2>&1; echo "rc=$?" >> $OUT/host_ubsan.log; tail -3 $OUT/host_ubsan.log
( timeout 600 python tools/gpu_stress.py 200 ) > $OUT/stress.log 2>&1; tail -2 $OUT/stress.log
( timeout 300 python tools/gpu_stag_spec_stress.py 200 5 ) > $OUT/stag_spec_stress.log 2>&1; tail -2 $OUT/stag_spec_stress.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-300 $OUT/bench.json
( timeout 300 python bench.py --feed host --steps 12 --warmup 3 --no-extras --no-cpu-baseline ) 2> /dev/null | grep '^{' > $OUT/bench_feed_host.json; cut -c1-200 $OUT/bench_feed_host.json
( timeout 300 python bench.py --feed jpeg --steps 12 --warmup 3 --no-extras --no-cpu-baseline ) 2> /dev/null | grep '^{' > $OUT/bench_feed_jpeg.json; cut -c1-200 $OUT/bench_feed_jpeg.json
rm -rf $OUT/prof; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-110
rm -rf $OUT/prof
STAG_CHILD=1 NOQ=1 CTX=256 B=1024 STEPS=2 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/profs -o r -- python tools/gpu_stag_batch.py > $OUT/profs.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profs -name '*.db' | head -1) > $OUT/stag_kernel_stats.csv; head -8 $OUT/stag_kernel_stats.csv | cut -c1-110; grep fps $OUT/profs.log | tail -1 | cut -c1-150
rm -rf $OUT/profs
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profj -o r -- python tools/gpu_jpeg_bench.py 256 80 > $OUT/profj.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profj -name '*.db' | head -1) > $OUT/jpeg_kernel_stats.csv; head -6 $OUT/jpeg_kernel_stats.csv | cut -c1-110
rm -rf $OUT/profj
bash tools/gpu_trace_stag.sh > $OUT/stag_single_trace.log 2>&1; tail -3 $OUT/stag_single_trace.log
bash tools/gpu_r6_stag_trace.sh 16 32 64 > $OUT/stag_group_sizes.txt 2>&1; tail -32 $OUT/stag_group_sizes.txt
( timeout 120 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
NO_REF=1 timeout 200 python tools/stag_bench.py > $OUT/stag_single.log 2>&1; tail -1 $OUT/stag_single.log
python - <<PY
import json, sys
sys.path.insert(0, "/root/repo")
from fiducials_amd import _lib
json.dump({"library_sha256": "$SHA", "device_text_sha256": _lib.device_text_sha256(),
           "files": ["r06_kernel_stats.csv", "r06_stag_kernel_stats.csv", "r06_jpeg_kernel_stats.csv", "r06_stag_single_trace.log", "r06_stag_group_sizes.txt"],
           "commands": {"r06_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras",
                        "r06_stag_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python tools/gpu_stag_batch.py (CTX=256 B=1024 STEPS=2, default hardware queues)",
                        "r06_jpeg_kernel_stats.csv": "rocprofv3 --kernel-trace --stats -- python tools/gpu_jpeg_bench.py 256 80",
                        "r06_stag_single_trace.log": "tools/gpu_trace_stag.sh (one cfg 5 frame queued ahead, kernel by kernel)",
                        "r06_stag_group_sizes.txt": "tools/gpu_r6_stag_trace.sh 16 32 64 (one STag group alone on the chip, kernel time per cycle by group size)"}},
          open("$OUT/kernel_stats.json", "w"), indent=1)
PY
echo $SHA | tee $OUT/lib.sha256
