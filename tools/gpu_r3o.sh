#!/bin/bash
# STag after a refine change: GPU tests, single-frame timeline (refine's duration), batch rate
mkdir -p gpurun_out/r3o
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
timeout 900 python -m pytest tests/test_gpu_stag.py -x -q -m gpu > gpurun_out/r3o/pytest.log 2>&1; tail -3 gpurun_out/r3o/pytest.log
bash tools/gpu_trace_stag.sh 2>&1 | grep "k_stag_refine\|span us\|k_stag_route_walk\|k_stag_pose"
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
STAG_CHILD=1 CTX=64 B=128 STEPS=4 timeout 300 python tools/gpu_stag_batch.py 2>&1 | tail -2 | cut -c1-300
