#!/bin/bash
# depth / chain point after the instruction cuts
mkdir -p gpurun_out/r3n
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
run() {  # depth chain_at
  FID_CHAIN_AT=$2 timeout 200 python bench.py --steps 30 --warmup 5 --in-flight $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $1 chain_at $2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'])"
}
run 2 0; run 3 0; run 4 0; run 2 1; run 3 1; run 4 1; run 3 2; run 3 0; run 2 0
