#!/bin/bash
# After a kernel change: the parity tests of the aruco path (or the pytest selection given as arguments), then the bench line
# twice with batches in turn and once one call after the other.   gpurun -- bash tools/gpu_check.sh [pytest args]
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out/check
sel=("$@"); [ ${#sel[@]} -eq 0 ] && sel=(tests/test_gpu_parity.py tests/test_gpu_pipeline.py)
timeout 900 python -m pytest "${sel[@]}" -x -q -m gpu > gpurun_out/check/pytest.log 2>&1; tail -3 gpurun_out/check/pytest.log
STEPS=20 bash tools/gpu_sweep.sh "" "" " -- --in-flight 1"
