#!/bin/bash
export TMPDIR=/tmp
for q in 4 16; do for sf in 128 86 64 43; do echo "== queues $q sub_frames $sf"; GPU_MAX_HW_QUEUES=$q FID_SUB_FRAMES=$sf python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c75-190; done; done
