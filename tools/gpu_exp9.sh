#!/bin/bash
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --timeout 300 ) 2>&1 | tail -2
timeout 600 python tools/gpu_ab.py "AB_TAG=default" "FID_SUB_FRAMES=256" 2>&1 | grep cfg | cut -c1-640
bash tools/gpu_pmc3.sh 2>&1 | grep "find_starts\|per_frame\|threshold"
