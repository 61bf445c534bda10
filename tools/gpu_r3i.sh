#!/bin/bash
# batches in turn on two / three contexts, the next one held back until the one before is past point FID_CHAIN_AT
mkdir -p gpurun_out/r3i
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3i/pytest.log 2>&1; tail -3 gpurun_out/r3i/pytest.log
run() {  # depth chain_at shares unordered
  local tag="d$1_c$2_s${3//,/-}_u$4"
  FID_CHAIN_AT=$2 FID_BENCH_UNORDERED=$4 ${3:+env FID_SUB_SHARES=$3} timeout 200 python bench.py --steps 30 --warmup 4 --in-flight $1 --no-extras --no-cpu-baseline > gpurun_out/r3i/$tag.json 2> gpurun_out/r3i/$tag.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r3i/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], (d.get("one_at_a_time") or {}).get("frames_per_s"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["config"]["markers_per_frame_found"])
except Exception as e:
    print("$tag", "failed", e); print(open("gpurun_out/r3i/$tag.err").read()[-600:])
P
}
if [ -n "$SWEEP2" ]; then
for rep in 1 2; do
for c in 0 1 2; do run 2 $c "100" 0; done
for c in 0 1 2; do run 3 $c "100" 0; done
run 4 1 "100" 0
run 2 0 "" 0
done
exit 0
fi
for c in 0 1 2 3 4 5; do run 2 $c "" 0; done
run 2 3 "" 1
run 3 3 "" 0
run 3 1 "" 0
for sh in "50,50" "100" "58,42" "70,30"; do run 2 3 "$sh" 0; run 2 1 "$sh" 0; done
run 1 3 "" 0
