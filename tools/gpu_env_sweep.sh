#!/bin/bash
# sweep one environment variable: gpu_env_sweep.sh VAR v1 v2 ...
export TMPDIR=/tmp
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], {k: round(v,2) for k, v in d['stage_ms_per_step'].items() if v > 0.8})"
done
