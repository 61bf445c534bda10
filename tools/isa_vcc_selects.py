#!/usr/bin/env python3
"""Static count, per kernel, of v_cndmask_b32 instructions that read the implicit VCC (e32 form) WITHOUT the VCC-writing compare
directly in front of them -- the form tools/valu_calib.hip measures at ~17 cycles per instruction on gfx950 (a second select on
the same VCC, or a select separated from its compare), against 2 - 4 cycles for the e64 form on an SGPR pair and for the
select right behind its compare.  Usage: isa_vcc_selects.py file.s [name-substring ...]"""
import collections
import re
import sys

cur = None
cnt = collections.defaultdict(collections.Counter)
prev_writes_vcc = False
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+|k_\w+):", line)
    if m:
        cur = m.group(1)
        prev_writes_vcc = False
        continue
    if cur is None:
        continue
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        if t.startswith(".Lfunc_end"):
            cur = None
        continue
    op = t.split()[0]
    args = t[len(op):]
    if op.startswith("v_"):
        cnt[cur]["valu"] += 1
    if op.startswith("v_cndmask_b32"):
        cnt[cur]["cndmask"] += 1
        if re.search(r"\bvcc\b", args):
            cnt[cur]["cndmask_vcc"] += 1
            if not prev_writes_vcc:
                cnt[cur]["cndmask_vcc_far"] += 1
    if op.startswith(("v_addc", "v_subb", "v_subbrev")) and re.search(r"\bvcc\b", args):
        cnt[cur]["carry_vcc"] += 1
    if op == "s_nop" or op == "s_waitcnt":
        continue  # (does not separate a compare from its select as far as the forwarding goes -- unknown; counted as adjacent)
    prev_writes_vcc = bool(re.match(r"v_cmp\w*_e32$", op) or (op.startswith("v_cmp") and args.strip().startswith("vcc")))
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1]["cndmask_vcc_far"]):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    if c["valu"] < 50:
        continue
    print(f"{k[:60]:60s} valu {c['valu']:6d} cndmask {c['cndmask']:5d} on vcc {c['cndmask_vcc']:5d} not behind its compare {c['cndmask_vcc_far']:5d}")
