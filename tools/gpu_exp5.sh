#!/bin/bash
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q --timeout 300 ) 2>&1 | tail -3
bash tools/gpu_trace1.sh
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
