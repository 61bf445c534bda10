#!/bin/bash
# the checker under AddressSanitizer + UndefinedBehaviorSanitizer.  Builds oracle/_asan/{liboracle,
# libjpeg_oracle,libstag_ref}.so (the last one = the REFERENCE's own STag sources compiled in place, only where /root/reference is
# mounted) and runs the CPU test suite against them.  ASan in a Python process needs its runtime preloaded; leak checking is off
# (the interpreter's own allocations).  halt_on_error=0 + log_path: every finding is listed, the run goes on.
cd "$(dirname "$0")/.."
set -u
make -C oracle -s asan || exit 1
[ -d /root/reference ] && make -C oracle -s _asan/libstag_ref.so
OUT=${1:-/tmp/fid_sanitizers}; rm -rf $OUT; mkdir -p $OUT
export ORACLE_SO_DIR=$PWD/oracle/_asan
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$OUT/asan:new_delete_type_mismatch=1:alloc_dealloc_mismatch=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$OUT/ubsan
python -m pytest tests -q -m "not gpu" -p no:cacheprovider -k "oracle or marker_gen or messages" 2>&1 | tail -3
unset LD_PRELOAD
python tools/sanitizer_summary.py $OUT
