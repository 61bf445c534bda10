#!/bin/bash
# kernel timeline (csv) of a few cfg 3 steps -> gpurun_out/r3f/trace.csv
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/r3f; mkdir -p gpurun_out/r3f
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3f/p -o r -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3f/log.txt 2>&1
tail -1 gpurun_out/r3f/log.txt | cut -c1-120
cp $(find gpurun_out/r3f/p -name '*kernel_trace.csv' | head -1) gpurun_out/r3f/trace.csv
rm -rf gpurun_out/r3f/p; ls -la gpurun_out/r3f
