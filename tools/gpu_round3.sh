#!/bin/bash
# rocprofv3 kernel stats of the JPEG ingest (256 frames per call) -> gpurun_out/jpeg_kernel_stats.csv
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/profj
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 32))" > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/profj -o r -- python tools/gpu_jpeg_bench.py 256 80 > gpurun_out/profj.log 2>&1
tail -1 gpurun_out/profj.log | cut -c1-400
python tools/rocpd_stats.py $(find gpurun_out/profj -name '*.db' | head -1) > gpurun_out/jpeg_kernel_stats.csv
head -12 gpurun_out/jpeg_kernel_stats.csv | cut -c1-150
