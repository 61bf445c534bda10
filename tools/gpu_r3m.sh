#!/bin/bash
# parity tests + the bench line (no extras) after a kernel change
mkdir -p gpurun_out/r3m
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3m/pytest.log 2>&1; tail -3 gpurun_out/r3m/pytest.log
for i in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r3m/bench_$i.json 2> gpurun_out/r3m/bench_$i.err
python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r3m/bench_$i.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["config"]["markers_per_frame_found"], json.dumps(d["stage_ms_per_step"]))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r3m/bench_$i.err").read()[-600:])
P
done
timeout 200 python bench.py --steps 20 --warmup 5 --in-flight 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one at a time', d['value'], json.dumps(d['stage_ms_per_step']))"
