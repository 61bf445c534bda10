#!/bin/bash
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out/r3c
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
for cfg in "${@}"; do
  echo "== dbg3 [$cfg]"
  FID_LIB=build_dbg/libfid_dbg3.so AB_CHILD=1 AB_STEPS=1 timeout 200 python tools/gpu_ab.py "$cfg" 2>&1 | grep -v "^resolve\|amdgpu.ids" | tail -4 | cut -c1-600
done
