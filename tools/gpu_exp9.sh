#!/bin/bash
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q --timeout 300 ) 2>&1 | tail -2
timeout 600 python tools/gpu_ab.py "AB_TAG=default" "FID_SUB_FRAMES=256" 2>&1 | grep cfg | cut -c1-640
