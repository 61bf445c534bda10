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
Static = every instruction counts once whatever its loop depth: a proxy for the dynamic mix, good enough to say which peak a
kernel's SQ_INSTS_VALU should be priced against.  Usage: isa_classes.py [lib.so] [calib.json] > profiles/r04_isa_classes.json"""
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

out = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "calibration": "profiles/r04_valu_calib.json",
       "classes": {"fast": sorted(fast), "slow4": sorted(slow4), "veryslow": sorted(veryslow)}, "kernels": {}}
cur = None
cnt = collections.defaultdict(collections.Counter)
unl = collections.Counter()
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        cur = m.group(1)
        continue
    if cur is None or "\t" not in line:
        continue
    text = line.split("//")[0].strip()
    if not text:
        continue
    op = text.split()[0]
    if op.startswith("v_"):
        c = classify(op, text)
        cnt[cur]["valu"] += 1
        cnt[cur][c] += 1
        if c == "unlisted":
            unl[re.sub(r"_(e32|e64)$", "", op)] += 1
    elif op.startswith("s_"):
        cnt[cur]["salu"] += 1
for k, c in cnt.items():
    if c["valu"] < 20:
        continue
    short = re.sub(r"^_Z\d+", "", k)
    short = re.match(r"[A-Za-z_0-9]+", short).group(0) if re.match(r"[A-Za-z_0-9]+", short) else k
    key = short if short not in out["kernels"] else k
    v = c["valu"]
    out["kernels"][key] = {"mangled": k, "valu": v, "salu": c["salu"], "fast": c["fast"], "slow4": c["slow4"] + c["unlisted"], "veryslow": c["veryslow"],
                           "unlisted": c["unlisted"], "fast_share": round(c["fast"] / v, 4)}
out["unlisted_opcodes"] = dict(unl.most_common(40))
json.dump(out, sys.stdout, indent=1)
