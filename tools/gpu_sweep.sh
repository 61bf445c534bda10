#!/bin/bash
export TMPDIR=/tmp
for v in 0 40000 80000 120000 160000; do
  echo "== FID_WALK_LDS=$v"
  FID_WALK_LDS=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], {k: v for k, v in d['stage_ms_per_step'].items() if 'walk' in k})"
done
