#!/bin/bash
export TMPDIR=/tmp FID_PROFILE=1
for sh in 4 3 2; do echo "== shift $sh"; FID_SEED_SHIFT=$sh timeout 200 python tools/gpu_latency.py 2>&1 | grep -v "^batch\|amdgpu.ids" | cut -c1-500; done
( timeout 400 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 ) 2>&1 | tail -3
