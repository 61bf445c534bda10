#!/bin/bash
# A/B/C... of several library builds on the cfg 3 batch (one context, one call after the other), fresh process each, in turn.
# Usage: gpu_ab_libs_n.sh rounds lib1.so lib2.so ...
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    FID_LIB=$L AB_CHILD=1 timeout 120 python tools/gpu_ab.py "" 2>&1 | grep fps | python -c "import sys,json; [print('$L', {k:d[k] for k in ('fps','ms_per_step')}, {k:d['stage_ms'][k] for k in ('threshold','seed_walk','approx','sort_cands','near','resolve','identify','subpix')}) for d in map(json.loads, sys.stdin)]"
  done
done
