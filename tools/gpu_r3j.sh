#!/bin/bash
# chained batches (two contexts, one piece each): knobs around the defaults
mkdir -p gpurun_out/r3j
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3j/pytest.log 2>&1; tail -3 gpurun_out/r3j/pytest.log
run() {  # tag, env...
  local tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r3j/$tag.json 2> gpurun_out/r3j/$tag.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r3j/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["config"]["markers_per_frame_found"])
except Exception as e:
    print("$tag", "failed", e); print(open("gpurun_out/r3j/$tag.err").read()[-600:])
P
}
run base A=1
run sw3 FID_SW_BLOCKS=3
run sw4 FID_SW_BLOCKS=4
run w2d1 FID_WALK2_DIV=1
run w2d4 FID_WALK2_DIV=4
run wb4 FID_WALK_BLOCKS=4
run wb6 FID_WALK_BLOCKS=6
run tg2k FID_TAIL_GRID=2048
run tg8k FID_TAIL_GRID=8192
run c1 FID_CHAIN_AT=1
run copy128 FID_COPY_BLOCKS=128
run copy512 FID_COPY_BLOCKS=512
run base2 A=1
( unset GPU_MAX_HW_QUEUES; run defq A=1 )
run q4 GPU_MAX_HW_QUEUES=4
