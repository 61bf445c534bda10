#!/usr/bin/env python3
"""VALU instructions of one kernel by source line (device assembly built with -gline-tables-only).
Usage: isa_lines.py file.s kernel-substring [top-n]"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
cur = None
loc = None
cnt = collections.Counter()
for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"^(_Z\w+|k_\w+):", line)
    if m:
        cur = m.group(1) if want in m.group(1) else None
        loc = None
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if t.startswith("v_"):
        cnt[loc] += 1
tot = sum(cnt.values())
print("total VALU", tot)
for (k, n) in cnt.most_common(top):
    print(f"{n:6d} {100.0 * n / tot:5.1f}%  {k}")
