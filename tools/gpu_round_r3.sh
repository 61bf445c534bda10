#!/bin/bash
# Round 3 evidence, part 1: full GPU test suite, host UBSan run, stress sweep, default bench line, kernel stats (aruco, STag, JPEG)
set -u
export TMPDIR=/tmp
cd /root/repo; OUT=gpurun_out/r3final; rm -rf $OUT; mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -q --timeout 400 ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
( FID_HOST_UBSAN=1 timeout 300 python -m pytest tests/test_gpu_host_cpp.py -q -m gpu ) > $OUT/host_ubsan.log 2>&1; echo "rc=$?" >> $OUT/host_ubsan.log; tail -3 $OUT/host_ubsan.log
( timeout 600 python tools/gpu_stress.py 200 ) > $OUT/stress.log 2>&1; tail -2 $OUT/stress.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-300 $OUT/bench.json
rm -rf $OUT/prof; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-110
rm -rf $OUT/prof
python -c "import bench; bench.make_stag_frames(bench.shard_seeds(0, 1, 16, 'stag'))" > /dev/null 2>&1
STAG_CHILD=1 CTX=64 B=128 STEPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/profs -o r -- python tools/gpu_stag_batch.py > $OUT/profs.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profs -name '*.db' | head -1) > $OUT/stag_kernel_stats.csv; head -8 $OUT/stag_kernel_stats.csv | cut -c1-110; tail -1 $OUT/profs.log | cut -c1-150
rm -rf $OUT/profs
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profj -o r -- python tools/gpu_jpeg_bench.py 256 80 > $OUT/profj.log 2>&1
python tools/rocpd_stats.py $(find $OUT/profj -name '*.db' | head -1) > $OUT/jpeg_kernel_stats.csv; head -6 $OUT/jpeg_kernel_stats.csv | cut -c1-110
rm -rf $OUT/profj
( timeout 120 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
sha256sum fiducials_amd/lib/libfid_amd.so | tee $OUT/lib.sha256
