#!/bin/bash
# Round 5, STag groups of 32 (routing context behind a pointer): parity, then the cfg 5 batch by slots / group size / queue setting
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r5stag; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_stag.py tests/test_gpu_host_cpp.py -m gpu -q -x --timeout 300 ) > $OUT/tests2.log 2>&1; tail -4 $OUT/tests2.log | cut -c1-250
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
NOQ=1 timeout 900 python tools/gpu_stag_batch.py "CTX=128 B=256 FID_VERBOSE=1" "CTX=128 B=256 FID_STAG_GROUP=16" "CTX=128 B=256 FID_STAG_GROUP=32" "CTX=64 B=256 FID_STAG_GROUP=32" "CTX=96 B=288 FID_STAG_GROUP=24" "CTX=192 B=384 FID_STAG_GROUP=32" "CTX=128 B=256 FID_STAG_GROUP=32 FID_STAG_SPEC=1" "CTX=128 B=256 FID_STAG_GROUP=32 FID_STAG_TILE_KB=24" "CTX=128 B=256 FID_STAG_GROUP=32 FID_STAG_TILE_KB=64" "CTX=128 B=256 FID_STAG_GROUP=32 GPU_MAX_HW_QUEUES=8" 2>&1 | cut -c1-420 | tee $OUT/batch2.log
