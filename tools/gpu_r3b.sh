#!/bin/bash
# round 3: parity of the tracing modes, then A/B of the cfg 3 batch (fresh process per configuration)
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out/r3b
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -25
for cfg in "${@}"; do
  AB_CHILD=1 timeout 200 python tools/gpu_ab.py "$cfg" 2>&1 | grep '^{' | cut -c1-1000
done
