#!/bin/bash
# A/B several builds of the library: prints fps and the walk stages
export TMPDIR=/tmp
for d in "$@"; do
  echo "== $d"
  FID_LIB=/root/repo/fiducials_amd/$d/libfid_amd.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], {k: round(v,2) for k, v in d['stage_ms_per_step'].items() if v > 0.2})"
done
