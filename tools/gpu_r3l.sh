#!/bin/bash
# single-frame latency in a fresh process (cfg 2): with and without the chain event
mkdir -p gpurun_out/r3l
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
for v in 0 100 0 100; do
FID_CHAIN_AT=$v timeout 200 python - <<P
import json, bench, torch
torch.cuda.init()
fr = bench.make_frames(bench.shard_seeds(0, 1, 1))
print("chain_at $v", json.dumps(bench.cfg2_latency(0, fr[0])))
P
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3l/lat.log
