#!/usr/bin/env python3
"""Static instruction counts per kernel from the device assembly (hipcc --cuda-device-only -S): VALU / SALU / LDS / VMEM lines
between a kernel's label and its end.  A proxy only (loops count once), good for before / after of straight-line code.
Usage: isa_count.py file.s [name-substring ...]"""
import collections
import re
import sys

cur = None
cnt = collections.defaultdict(lambda: collections.Counter())
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+|k_\w+):", line)
    if m:
        cur = m.group(1)
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith(".Lfunc_end"):
        cur = None
        continue
    op = t.split(" ")[0].split("\t")[0]
    if op.startswith("v_"):
        cnt[cur]["valu"] += 1
        if op.startswith(("v_writelane", "v_readlane", "v_readfirstlane")):
            cnt[cur]["lane"] += 1
    elif op.startswith("s_"):
        cnt[cur]["salu"] += 1
    elif op.startswith("ds_"):
        cnt[cur]["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cnt[cur]["vmem"] += 1
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1]["valu"]):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(f"{k[:70]:70s} valu {c['valu']:6d} salu {c['salu']:6d} lds {c['lds']:5d} vmem {c['vmem']:5d} lane {c['lane']:4d}")
