#!/bin/bash
export TMPDIR=/tmp
( timeout 800 python -m pytest tests/test_gpu_stag.py -x -q --timeout 300 ) 2>&1 | tail -3
python bench.py --workload stag --streams 16 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200
NO_REF=1 python tools/stag_bench.py 2>&1 | grep GPU
bash tools/gpu_trace_stag.sh 2>&1 | grep "refine\|route_walk\|span"
