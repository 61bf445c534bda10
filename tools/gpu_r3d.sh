#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of bench.py under the given environment settings, one run per argument,
# then the SQ counter pass of the last one
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out/r3d
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
i=0
for cfg in "${@}"; do
  i=$((i+1)); rm -rf gpurun_out/r3d/p$i
  echo "== [$cfg]"
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3d/p$i -o r -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3d/p$i.log 2>&1
  tail -1 gpurun_out/r3d/p$i.log | cut -c1-200
  python tools/rocpd_stats.py $(find gpurun_out/r3d/p$i -name '*.db' | head -1) > gpurun_out/r3d/stats$i.csv
  cut -d, -f1-4 gpurun_out/r3d/stats$i.csv | sed 's/(.*)",/",/' | head -24
  find gpurun_out/r3d/p$i -name '*.db' -delete
done
