#!/bin/bash
# A/B of configurations through bench.py (no extras, no CPU baseline), a fresh process each: one result line per argument.
# An argument is "ENV=VALUE ENV=VALUE ... [-- bench flags]", e.g.
#   gpurun -- bash tools/gpu_sweep.sh "" "FID_CHAIN_AT=1" "FID_SUB_SHARES=50,50 -- --in-flight 3" "GPU_MAX_HW_QUEUES=4"
# STEPS / WARMUP (default 30 / 5) apply to every line.  (The library's knobs: getenv calls in fiducials_amd/csrc/fid_api.hip.)
export TMPDIR=/tmp
cd /root/repo
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 256))" > /dev/null 2>&1
for cfg in "${@:-}"; do
  envs="${cfg%%--*}"; flags=""; [[ "$cfg" == *--* ]] && flags="${cfg#*--}"
  env $envs timeout 300 python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-5} --no-extras --no-cpu-baseline $flags 2>/dev/null | python -c "
import sys, json
cfg = sys.argv[1]
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    r = d['roofline']
    print(f'[{cfg}] {d[\"value\"]:.0f} frames/s, {d[\"ms_per_step\"]} ms/step, in flight {d[\"config\"][\"in_flight\"]}, markers/frame {d[\"config\"][\"markers_per_frame_found\"]}; '
          f'{r[\"kernel\"]} {r[\"kernel_ms_per_launch\"]} ms ({r[\"frac\"]}); ' + ' '.join(f'{k}={v:.2f}' for k, v in d['stage_ms_per_step'].items() if v > 0.25))
except Exception as e:
    print(f'[{cfg}] failed: {e!r}')
" "$cfg"
done
