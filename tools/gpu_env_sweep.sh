#!/bin/bash
# sweep: each argument is a full "VAR=val VAR2=val" environment string
export TMPDIR=/tmp
for e in "$@"; do
  echo "== $e"
  env $e timeout 200 python bench.py --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], {k: round(v,2) for k, v in d['stage_ms_per_step'].items() if v > 0.6})"
done
