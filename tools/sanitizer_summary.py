#!/usr/bin/env python3
"""Summarise the ASan / UBSan logs of tools/run_sanitizers.sh: findings by kind and by the source file of the first frame
that lies in the reference tree (REF:) or in this repository."""
import collections
import glob
import re
import sys

errs, own = collections.Counter(), collections.Counter()
for f in glob.glob(sys.argv[1] + "/*"):
    txt = open(f, errors="replace").read()
    for blk in re.split(r"(?==+\d+==ERROR)|(?=\S+: runtime error)", txt):
        m = re.search(r"ERROR: AddressSanitizer: (\S+)|(runtime error: [^\n]*)", blk)
        if not m:
            continue
        kind = m.group(1) or m.group(2)
        frames = re.findall(r"#\d+ 0x[0-9a-f]+ in (\S+) (/root/\S+)", blk)
        user = [(fn, loc) for fn, loc in frames if "/root/reference" in loc or "/root/repo" in loc]
        top = user[0] if user else ("?", "?")
        errs[(kind, top[1].split(":")[0].replace("/root/reference/stag_detect/", "REF:").replace("/root/repo/", ""))] += 1
        if "/root/repo" in top[1]:
            own[(kind, top[0], top[1].replace("/root/repo/", ""))] += 1
print("findings by (kind, file of the first reference / repository frame):")
for k, v in errs.most_common(60):
    print(f"{v:6d}  {k[0]:45s} {k[1]}")
if not errs:
    print("     0")
print("first frame in this repository's sources (call sites of reference code inlined from its headers: EdgeMap / EDLines destructors):")
for k, v in own.most_common():
    print(f"{v:6d}  {k[0]:30s} {k[1]} {k[2]}")
if not own:
    print("     0")
