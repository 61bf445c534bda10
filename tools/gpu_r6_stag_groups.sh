#!/bin/bash
# Round 6: frames per merged launch.  The cfg 5 batch at group sizes 16 / 32 / 64 (and more slots), default hardware queues, the
# verbose line (launches recorded / issued / unmergeable, host time) beside each; STag GPU tests first.
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r6groups; rm -rf $OUT; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_stag.py -m gpu -q -x --timeout 400 ) > $OUT/pytest_stag.log 2>&1; tail -5 $OUT/pytest_stag.log
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
for round in 1 2; do
FID_VERBOSE=1 NOQ=1 python tools/gpu_stag_batch.py "CTX=128 B=256 FID_STAG_GROUP=16" "CTX=128 B=256 FID_STAG_GROUP=32" "CTX=128 B=256 FID_STAG_GROUP=64" \
   "CTX=128 B=512 FID_STAG_GROUP=64" "CTX=192 B=384 FID_STAG_GROUP=64" "CTX=256 B=512 FID_STAG_GROUP=64" "CTX=64 B=256 FID_STAG_GROUP=64" "CTX=64 B=256 FID_STAG_GROUP=32" 2>&1 | cut -c1-700 | tee -a $OUT/groups.log
done
