#!/bin/bash
# kernel timeline / stats of fid_stag_detect_markers_batch in group mode -> gpurun_out/stagb/
export TMPDIR=/tmp
cd /root/repo; rm -rf gpurun_out/stagb; mkdir -p gpurun_out/stagb
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
env STAG_CHILD=1 STEPS=2 "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/stagb/p -o r -- python tools/gpu_stag_batch.py > gpurun_out/stagb/log.txt 2>&1
tail -2 gpurun_out/stagb/log.txt | cut -c1-300
cp $(find gpurun_out/stagb/p -name '*kernel_trace.csv' | head -1) gpurun_out/stagb/trace.csv
cp $(find gpurun_out/stagb/p -name '*memory_copy_trace.csv' | head -1) gpurun_out/stagb/copies.csv 2>/dev/null
rm -rf gpurun_out/stagb/p; ls -la gpurun_out/stagb
