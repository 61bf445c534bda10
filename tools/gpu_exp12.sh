#!/bin/bash
export TMPDIR=/tmp
for cfg in "24 20" "24 24" "28 24" "24 12"; do set -- $cfg; echo "== GPU_MAX_HW_QUEUES=$1 contexts $2"; GPU_MAX_HW_QUEUES=$1 python bench.py --workload stag --streams $2 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200; done
