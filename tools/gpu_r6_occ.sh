#!/bin/bash
# Round 6, the look that comes first: what is RESIDENT while the STag batch (cfg 5, groups of 16 on the runtime's default queues)
# and the aruco bench (cfg 3, two contexts) run.  Kernel traces as CSV (per dispatch: block, grid, LDS, registers, start / end) for
# tools/occupancy.py, one SQ counter pass each for the measured side (waves, busy-CU cycles, wave cycles), the aruco single-frame
# timeline (cfg 2).  -> gpurun_out/r6occ/
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r6occ; rm -rf $OUT; mkdir -p $OUT
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag')); bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
# --- STag batch, 128 slots / 256 frames (the bench's cfg 5 shape), default queues
STAG_CHILD=1 NOQ=1 CTX=128 B=256 STEPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/st -o r -- python tools/gpu_stag_batch.py > $OUT/stag_trace.log 2>&1
cp $(find $OUT/st -name '*kernel_trace.csv' | head -1) $OUT/stag_batch_trace.csv; rm -rf $OUT/st; tail -1 $OUT/stag_trace.log | cut -c1-160
STAG_CHILD=1 NOQ=1 CTX=128 B=256 STEPS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES --output-format csv -d $OUT/stc -o r -- python tools/gpu_stag_batch.py > $OUT/stag_pmc.log 2>&1
mkdir -p $OUT/stag_counters; cp $(find $OUT/stc -name '*counter_collection.csv' | head -1) $OUT/stag_counters/counters.csv 2>/dev/null; rm -rf $OUT/stc; tail -2 $OUT/stag_pmc.log | cut -c1-200
# (a group on its own: one group of 16 frames, nothing beside it -- what a kernel of the group does to the chip alone)
STAG_CHILD=1 NOQ=1 CTX=16 B=32 STEPS=2 FID_STAG_GROUP=16 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/st1 -o r -- python tools/gpu_stag_batch.py > $OUT/stag_trace1.log 2>&1
cp $(find $OUT/st1 -name '*kernel_trace.csv' | head -1) $OUT/stag_group_alone_trace.csv; rm -rf $OUT/st1; tail -1 $OUT/stag_trace1.log | cut -c1-160
# --- aruco bench, two contexts in turn
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ar -o r -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/aruco_trace.log 2>&1
cp $(find $OUT/ar -name '*kernel_trace.csv' | head -1) $OUT/aruco_bench_trace.csv; rm -rf $OUT/ar; tail -1 $OUT/aruco_trace.log | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES --output-format csv -d $OUT/arc -o r -- python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/aruco_pmc.log 2>&1
mkdir -p $OUT/aruco_counters; cp $(find $OUT/arc -name '*counter_collection.csv' | head -1) $OUT/aruco_counters/counters.csv 2>/dev/null; rm -rf $OUT/arc; tail -2 $OUT/aruco_pmc.log | cut -c1-200
# --- the aruco single frame (cfg 2), kernel by kernel
bash tools/gpu_trace1.sh > $OUT/aruco_single_trace.log 2>&1; tail -40 $OUT/aruco_single_trace.log
# --- the sheet
python tools/occupancy.py --label "round 6, library as built at the start of the round" --trace $OUT/stag_batch_trace.csv $OUT/aruco_bench_trace.csv $OUT/stag_group_alone_trace.csv \
   --check $OUT/stag_counters/counters.csv $OUT/aruco_counters/counters.csv > $OUT/occupancy.json 2> $OUT/occupancy.err; tail -3 $OUT/occupancy.err
gzip -9 $OUT/*_trace.csv; ls -la $OUT
# --- and that the round's first changes (advice items) left the STag road where it was
( time timeout 900 python -m pytest tests/test_gpu_stag.py tests/test_overlay.py -m gpu -q -x --timeout 400 ) > $OUT/pytest_stag.log 2>&1; tail -5 $OUT/pytest_stag.log
( timeout 300 python tools/gpu_stag_spec_stress.py 120 3 ) > $OUT/spec_stress.log 2>&1; tail -4 $OUT/spec_stress.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-400 $OUT/bench.json
