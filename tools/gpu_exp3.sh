#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/gpu_ab.py "" "" "" "FID_THR=tile" "" 2>&1 | grep cfg | cut -c1-200
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
