#!/bin/bash
export TMPDIR=/tmp
for k in 4 12 16 24; do FID_LIB=/root/repo/build_dbg/libfid_ck$k.so timeout 300 python tools/gpu_ab.py "AB_TAG=ckpt$k" 2>&1 | grep cfg | cut -c1-640; done
timeout 300 python tools/gpu_ab.py "AB_TAG=ckpt8" 2>&1 | grep cfg | cut -c1-640
