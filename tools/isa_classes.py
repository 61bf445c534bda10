#!/usr/bin/env python3
"""Static VALU issue-class mix of every kernel in the SHIPPED library: the gfx950 code object is taken out of
fiducials_amd/lib/libfid_amd.so (.hip_fatbin -> clang-offload-bundler), disassembled with llvm-objdump, and every VALU
instruction is put into the issue class tools/valu_calib.hip MEASURED for its opcode (profiles/r04_valu_calib.json):
  2-cycle class  ~ 900 - 1050 G wave-instructions/s on the chip (v_add/sub_u32, v_and/or/xor/not, v_lshrrev/ashrrev, v_mov,
                   v_add/sub/mul/fma_f32)
  4-cycle class  ~ 570 - 590 G/s (everything else that was measured: v_mul_*24, v_mad_*, v_dot2c, v_pk_*, v_alignbit, v_bfe,
                   v_lshlrev, min / max, every three-operand integer op, conversions, bit counts, DPP forms, v_cmp, v_cndmask,
                   v_readlane / v_writelane, f64 add / mul / fma)
  slow           v_rcp/sqrt_f64 and friends (quarter of that again)
Opcodes the calibration does not list are put into the 4-cycle class and counted as `unlisted`.
`fast_share` is static: every instruction counts once whatever its loop depth.  Round 5 adds the proxy the round-4 review asked for
-- `fast_share_loops`: the same share over the instructions that lie INSIDE a loop only (a backward branch to an earlier address
of the same kernel closes a loop; prologues, epilogues and straight-line set-up do not count), and `fast_share_depth`: every
instruction weighted by 8 ^ (loop depth) -- what a kernel executes most is what its SQ_INSTS_VALU should be priced by.  bench.py
uses fast_share_loops where a kernel has loops.  Usage: isa_classes.py [lib.so] [calib.json] > profiles/isa_classes.json"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fiducials_amd", "lib", "libfid_amd.so")
calib = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_valu_calib.json")))

# opcode -> class from the measurement (chip rate at 8 waves per SIMD)
fast, slow4, veryslow = set(), set(), set()
for name, v in calib["kinds"].items():
    if v["counts"] != "VALU" or not name.startswith("v_"):
        continue
    op = name.split()[0]
    rate = v["by_waves_per_simd"]["8"]["chip_ginstr_s"]
    if "dpp" in name or "cndmask" in name or "cmp" in name:
        continue  # forms, not opcodes: handled below
    (fast if rate > 800 else (slow4 if rate > 300 else veryslow)).add(op)


def classify(op: str, text: str) -> str:
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if "row_" in text or "quad_perm" in text or "wave_" in text or "_dpp" in op or "_sdwa" in op:
        return "slow4"  # DPP / SDWA forms measured in the 4-cycle class even for v_add_u32 / v_mov_b32
    if base in fast:
        return "fast"
    if base in veryslow or re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos|frexp|ldexp|trig|div_fixup|div_fmas|div_scale)", base) and "f64" in base:
        return "veryslow"
    if base in slow4:
        return "slow4"
    if base.startswith(("v_cmp", "v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane")):
        return "slow4"
    return "unlisted"


with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
    # (an output file is named on purpose: llvm-objcopy with one file argument rewrites its INPUT in place)
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "copy.so")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout

sys.path.insert(0, ROOT)
from fiducials_amd import _lib as _fl  # noqa: E402

out = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "device_text_sha256": _fl.device_text_sha256(lib), "calibration": "profiles/r04_valu_calib.json",
       "classes": {"fast": sorted(fast), "slow4": sorted(slow4), "veryslow": sorted(veryslow)}, "kernels": {}}
cur = None
cnt = collections.defaultdict(collections.Counter)
unl = collections.Counter()
insts = collections.defaultdict(list)   # kernel -> [(address, class or None)]
loops = collections.defaultdict(list)   # kernel -> [(first address, last address)] of every backward branch
base_addr = {}
for line in dis.splitlines():
    m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
    if m:
        cur = m.group(2)
        base_addr[cur] = int(m.group(1), 16)
        continue
    if cur is None or "\t" not in line:
        continue
    text = line.split("//")[0].strip()
    if not text:
        continue
    am = re.search(r"//\s*([0-9A-Fa-f]+):", line)
    addr = int(am.group(1), 16) if am else None
    op = text.split()[0]
    c = None
    if op.startswith("v_"):
        c = classify(op, text)
        cnt[cur]["valu"] += 1
        cnt[cur][c] += 1
        if c == "unlisted":
            unl[re.sub(r"_(e32|e64)$", "", op)] += 1
    elif op.startswith("s_"):
        cnt[cur]["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")) and addr is not None:
            tm = re.search(r"<(\S+?)\+0x([0-9a-f]+)>\s*$", line)
            if tm and tm.group(1) == cur:
                tgt = base_addr[cur] + int(tm.group(2), 16)
                if tgt <= addr:
                    loops[cur].append((tgt, addr))
    if addr is not None:
        insts[cur].append((addr, c))
for k in list(cnt):
    lp = loops.get(k, [])
    inl = collections.Counter()
    wsum = collections.Counter()
    for addr, c in insts[k]:
        if c is None:
            continue
        depth = sum(1 for a, b in lp if a <= addr <= b)
        cc = "fast" if c == "fast" else "other"
        if depth > 0:
            inl[cc] += 1
        wsum[cc] += 8 ** min(depth, 6)
    cnt[k]["loop_valu"] = inl["fast"] + inl["other"]
    cnt[k]["loop_fast"] = inl["fast"]
    cnt[k]["w_fast"] = wsum["fast"]
    cnt[k]["w_all"] = wsum["fast"] + wsum["other"]
    cnt[k]["nloops"] = len(lp)
for k, c in cnt.items():
    if c["valu"] < 20:
        continue
    short = re.sub(r"^_Z\d+", "", k)
    short = re.match(r"[A-Za-z_0-9]+", short).group(0) if re.match(r"[A-Za-z_0-9]+", short) else k
    key = short if short not in out["kernels"] else k
    v = c["valu"]
    out["kernels"][key] = {"mangled": k, "valu": v, "salu": c["salu"], "fast": c["fast"], "slow4": c["slow4"] + c["unlisted"], "veryslow": c["veryslow"],
                           "unlisted": c["unlisted"], "fast_share": round(c["fast"] / v, 4), "loops": c["nloops"], "valu_in_loops": c["loop_valu"],
                           "fast_share_loops": round(c["loop_fast"] / c["loop_valu"], 4) if c["loop_valu"] else None,
                           "fast_share_depth": round(c["w_fast"] / c["w_all"], 4) if c["w_all"] else None}
out["unlisted_opcodes"] = dict(unl.most_common(40))
json.dump(out, sys.stdout, indent=1)
