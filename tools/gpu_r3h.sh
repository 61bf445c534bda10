#!/bin/bash
# submit / collect: parity test, then the bench line at 1..4 batches in flight
mkdir -p gpurun_out/r3h
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3h/pytest.log 2>&1; tail -3 gpurun_out/r3h/pytest.log
for d in 1 2 3 4 3; do
  timeout 200 python bench.py --steps 24 --warmup 3 --in-flight $d --no-extras --no-cpu-baseline > gpurun_out/r3h/bench_$d.json 2> gpurun_out/r3h/bench_$d.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r3h/bench_$d.json").read().strip().splitlines()[-1])
    print($d, d["value"], d["ms_per_step"], d.get("one_at_a_time"), d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"])
except Exception as e:
    print($d, "failed", e); print(open("gpurun_out/r3h/bench_$d.err").read()[-800:])
P
done
(unset GPU_MAX_HW_QUEUES; timeout 200 python bench.py --steps 24 --warmup 3 --in-flight 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default queues', d['value'], d.get('one_at_a_time'))")
