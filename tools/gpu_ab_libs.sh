#!/bin/bash
# A/B of two library builds on the cfg 3 batch, fresh process each, alternating.  Usage: gpu_ab_libs.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in $A $B; do
    echo "== $L"
    FID_LIB=$L AB_CHILD=1 timeout 120 python tools/gpu_ab.py "" 2>&1 | grep fps | python -c "import sys,json; [print({k:d[k] for k in ('fps','ms_per_step')}, {k:d['stage_ms'][k] for k in ('threshold','seed_walk','approx','sort_cands','near','resolve','identify','subpix')}) for d in map(json.loads, sys.stdin)]"
  done
done
