#!/bin/bash
# round 3, call A: walker statistics of the seed walk / the survivor walk (FID_DEBUG_STATS builds), baseline stage times, SQ counters
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out/r3a
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
for m in 1 2; do
  echo "== dbg mode $m"
  FID_LIB=build_dbg/libfid_dbg$m.so AB_CHILD=1 AB_STEPS=1 timeout 200 python tools/gpu_ab.py "" > gpurun_out/r3a/dbg$m.log 2>&1
  grep -v "^resolve" gpurun_out/r3a/dbg$m.log | tail -14 | cut -c1-400
done
echo "== baseline"
AB_CHILD=1 timeout 200 python tools/gpu_ab.py "" 2>&1 | tail -2 | cut -c1-900
bash tools/gpu_pmc_sq.sh > gpurun_out/r3a/sq.txt 2>&1; tail -60 gpurun_out/r3a/sq.txt | cut -c1-330
