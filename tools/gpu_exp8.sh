#!/bin/bash
export TMPDIR=/tmp
timeout 900 python tools/gpu_ab.py "AB_TAG=default" "FID_WALK_BLOCKS=12" "FID_WALK_BLOCKS=16" "FID_WALK_BLOCKS=6" "FID_SUB_FRAMES=256 FID_WALK_BLOCKS=8" "FID_SUB_FRAMES=86" "FID_SUB_FRAMES=86 FID_WALK_BLOCKS=8" 2>&1 | grep cfg | cut -c1-640
