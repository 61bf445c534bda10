#!/bin/bash
# A/B of environment configurations on the cfg 3 batch, fresh process each (no tests)
export TMPDIR=/tmp
cd /root/repo
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
for cfg in "${@}"; do
  env $cfg AB_CHILD=1 timeout 200 python tools/gpu_ab.py "$cfg" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); s = d.get('stage_ms', {})
    print(d.get('cfg'), '| fps', d.get('fps'), 'same', d.get('same_as_first'), d.get('error', ''), '|', ' '.join(f'{k}={v:.2f}' for k, v in s.items() if v > 0.25))
"
done
