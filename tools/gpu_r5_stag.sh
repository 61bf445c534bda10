#!/bin/bash
# Round 5, STag: parity tests, then frames queued ahead (FID_STAG_SPEC, default on) against the counted road -- single frame and batch
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r5stag; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_stag.py tests/test_gpu_pipeline.py tests/test_gpu_host_cpp.py -m gpu -q -x --timeout 300 ) > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log | cut -c1-250
for s in 0 1; do
  echo "== FID_STAG_SPEC=$s single frame"; FID_STAG_SPEC=$s NO_REF=1 timeout 200 python tools/stag_bench.py 2>&1 | tail -2
done
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
NOQ=1 timeout 600 python tools/gpu_stag_batch.py "CTX=128 B=256 FID_STAG_SPEC=0 FID_VERBOSE=1" "CTX=128 B=256 FID_STAG_SPEC=1 FID_VERBOSE=1" "CTX=64 B=256 FID_STAG_SPEC=0" "CTX=64 B=256 FID_STAG_SPEC=1" "CTX=128 B=256 FID_STAG_SPEC=0" "CTX=128 B=256 FID_STAG_SPEC=1" 2>&1 | cut -c1-600 | tee $OUT/batch.log
