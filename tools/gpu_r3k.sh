#!/bin/bash
# streams of host-fed and of JPEG batches through two contexts in turn
mkdir -p gpurun_out/r3k
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
timeout 400 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3k/pytest.log 2>&1; tail -3 gpurun_out/r3k/pytest.log
timeout 500 python - > gpurun_out/r3k/legs.log 2>&1 <<'P'
import json, numpy as np, torch, bench
from fiducials_amd.synth import K_DEFAULT
torch.cuda.init()
frames = bench.make_frames(bench.shard_seeds(0, 1, 256))
print(json.dumps(bench.host_feed_result(0, frames, K_DEFAULT.copy(), np.zeros(5))))
r = bench.jpeg_side_result(0, frames)
print(json.dumps({k: r.get(k) for k in ("value", "jpeg_to_markers", "jpeg_stream_to_markers")}))
P
cut -c1-1200 gpurun_out/r3k/legs.log | tail -6
