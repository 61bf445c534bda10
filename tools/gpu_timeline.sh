#!/bin/bash
# kernel timeline (csv) of a few cfg 3 steps (batches in turn on two contexts) -> gpurun_out/timeline/trace.csv
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/timeline; mkdir -p gpurun_out/timeline
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline/p -o r -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/timeline/log.txt 2>&1
tail -1 gpurun_out/timeline/log.txt | cut -c1-120
cp $(find gpurun_out/timeline/p -name '*kernel_trace.csv' | head -1) gpurun_out/timeline/trace.csv
rm -rf gpurun_out/timeline/p; ls -la gpurun_out/timeline
