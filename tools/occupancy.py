#!/usr/bin/env python3
"""Occupancy sheet of the SHIPPED library (round 6): what of each kernel can be RESIDENT on a CU, and which kernels sit beside
which while the bench runs.

Static half (runs anywhere, no GPU): the gfx950 code object is taken out of fiducials_amd/lib/libfid_amd.so (.hip_fatbin ->
clang-offload-bundler) and its metadata note read with llvm-readelf --notes: per kernel .vgpr_count, .agpr_count, .sgpr_count,
.group_segment_fixed_size (static LDS), .max_flat_workgroup_size, .private_segment_fixed_size (scratch).

Dynamic half (optional, `--trace <kernel_trace.csv> [...]`): rocprofv3 --kernel-trace CSVs carry, per DISPATCH, Workgroup_Size,
Grid_Size, LDS_Block_Size (static + dynamic, what the launch asked for) and the start / end timestamps.  From them:
  * per kernel the launch shapes that occurred (block size, LDS per workgroup, workgroups per launch);
  * residency per CU by each resource, MI355X_MICROARCH.md "Register files" / "Residency":
        waves/SIMD by registers = min(8, floor(512 / (ceil((vgpr + agpr) / 8) * 8)))
        workgroups/CU by LDS    = floor(163840 / LDS per workgroup)         (LDS allocation granule: 512 B assumed here)
        workgroups/CU by waves  = floor(32 / waves per workgroup)           (32 waves per CU, 8 per SIMD)
        workgroups/CU by SGPRs  = floor(800 / (ceil(sgpr / 16) * 16 + 16)) * 4 / waves per workgroup   (per SIMD: the guide's rule)
    and the binding one; resident waves/CU = workgroups/CU x waves per workgroup;
  * `slots`: workgroups of ONE launch / (workgroups/CU x 256 CUs) -- below 1 the launch never fills the chip, above it the
    launch runs in that many rounds;
  * a co-residency matrix: for every pair of kernel names the time (ms) during which dispatches of both were in flight, and for
    each kernel the share of its in-flight time spent beside every other one -- which LDS / register footprints actually meet.
`--check <counter_collection.csv>`: a measured pass (SQ_WAVES, SQ_BUSY_CU_CYCLES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE, and
SQ_LEVEL_WAVES where the counter exists) next to the sheet: mean resident waves per busy CU = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES
(both summed over the SEs, x4 sampling cancels) against the sheet's waves/CU.

Usage: occupancy.py [--lib lib.so] [--trace a.csv [b.csv ...]] [--check counters.csv] [--label name] > profiles/r06_occupancy.json"""
import argparse
import collections
import csv
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LDS_PER_CU = 160 * 1024
CUS = 256


def short(name: str) -> str:
    """k_stag_batch<k_stag_route_walk_fn>(...) -> k_stag_route_walk[g]; template arguments kept, parameter list dropped."""
    n = name.replace("void ", "").strip()
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"k_stag_batch<(k_stag_\w+)_fn>", n)
    if m:
        return m.group(1) + "[g]"
    return re.sub(r"\s+", "", n)


def code_object(lib: str, td: str) -> str:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
    return co


def read_notes(co: str) -> dict:
    import yaml

    txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    body = txt[txt.index("---"):]
    body = body[: body.index("\n...")] if "\n..." in body else body
    meta = yaml.safe_load(body)
    out = {}
    for k in meta["amdhsa.kernels"]:
        out[k[".name"]] = {f: int(k.get("." + f, 0)) for f in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size",
                                                               "private_segment_fixed_size", "max_flat_workgroup_size", "vgpr_spill_count",
                                                               "sgpr_spill_count")}
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, p.stdout.splitlines()))


def residency(vgpr, agpr, sgpr, lds, block):
    waves_wg = max(1, (block + 63) // 64)
    alloc = max(8, -(-(vgpr + agpr) // 8) * 8)
    by_vgpr_simd = min(8, 512 // alloc)
    by_sgpr_simd = min(8, 800 // (-(-max(sgpr, 1) // 16) * 16 + 16))
    simd_waves = min(by_vgpr_simd, by_sgpr_simd)
    wg_by_regs = simd_waves * 4 // waves_wg if waves_wg <= 4 else simd_waves // (-(-waves_wg // 4))
    wg_by_waves = 32 // waves_wg
    lds_alloc = -(-lds // 512) * 512 if lds > 0 else 0
    wg_by_lds = LDS_PER_CU // lds_alloc if lds_alloc else 10 ** 6
    wg = max(0, min(wg_by_regs, wg_by_waves, wg_by_lds))
    bind = "lds" if wg == wg_by_lds and wg_by_lds < min(wg_by_regs, wg_by_waves) else ("registers" if wg == wg_by_regs and wg_by_regs < wg_by_waves else "wave slots")
    return {"waves_per_wg": waves_wg, "waves_per_simd_by_vgpr": by_vgpr_simd, "waves_per_simd_by_sgpr": by_sgpr_simd, "wg_per_cu_by_regs": wg_by_regs,
            "wg_per_cu_by_waves": wg_by_waves, "wg_per_cu_by_lds": None if wg_by_lds >= 10 ** 6 else wg_by_lds, "wg_per_cu": wg,
            "waves_per_cu": wg * waves_wg, "binding": bind, "lds_in_use_per_cu": wg * lds_alloc}


def fits_beside(host, nhost, guest):
    """How many workgroups of `guest` fit on a CU that already holds `nhost` workgroups of `host`?  Each = dict(vgpr, agpr, sgpr, lds,
    block).  Registers are counted per SIMD (a workgroup's waves spread over the four SIMDs round robin), LDS and wave slots per CU."""
    def alloc(k):
        return max(8, -(-(k["vgpr"] + k.get("agpr", 0)) // 8) * 8)

    def ldsa(k):
        return -(-k["lds"] // 512) * 512 if k["lds"] > 0 else 0

    hw, gw = max(1, (host["block"] + 63) // 64), max(1, (guest["block"] + 63) // 64)
    # waves per SIMD of the host: nhost workgroups x hw waves over four SIMDs
    host_waves_simd = -(-nhost * hw // 4)
    free_regs = 512 - host_waves_simd * alloc(host)
    free_lds = LDS_PER_CU - nhost * ldsa(host)
    free_waves = 32 - nhost * hw
    guest_waves_simd = max(0, min(free_regs // alloc(guest), 8 - host_waves_simd))
    by_regs = guest_waves_simd * 4 // gw
    by_lds = free_lds // ldsa(guest) if ldsa(guest) else 10 ** 6
    by_waves = free_waves // gw
    n = max(0, min(by_regs, by_lds, by_waves))
    return {"workgroups": n, "waves": n * gw, "by_registers": by_regs, "by_lds": None if by_lds >= 10 ** 6 else by_lds, "by_wave_slots": by_waves,
            "free_lds_bytes": free_lds, "free_vgprs_per_simd_lane": free_regs}


def read_trace(paths):
    disp = []
    for p in paths:
        with open(p) as fh:
            for r in csv.DictReader(fh):
                try:
                    bx = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
                    gx = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
                    disp.append({"name": short(r["Kernel_Name"]), "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]), "block": bx,
                                 "wgs": gx // max(bx, 1), "lds": int(r.get("LDS_Block_Size", 0) or 0), "vgpr": int(r.get("VGPR_Count", 0) or 0),
                                 "agpr": int(r.get("Accum_VGPR_Count", 0) or 0), "sgpr": int(r.get("SGPR_Count", 0) or 0), "file": os.path.basename(p)})
                except (KeyError, ValueError):
                    continue
    return disp


def coresidency(disp, top=14):
    """Sweep over dispatch start / end events: time during which >= 1 dispatch of A and >= 1 of B are in flight."""
    by_file = collections.defaultdict(list)
    for d in disp:
        by_file[d["file"]].append(d)
    alone = collections.Counter()
    pair = collections.Counter()
    busy = collections.Counter()
    for ds in by_file.values():
        ev = []
        for d in ds:
            ev.append((d["t0"], 1, d["name"]))
            ev.append((d["t1"], -1, d["name"]))
        ev.sort(key=lambda e: (e[0], e[1]))
        live = collections.Counter()
        last = ev[0][0] if ev else 0
        for t, s, n in ev:
            dt = t - last
            if dt > 0 and live:
                names = sorted(k for k, v in live.items() if v > 0)
                for a in names:
                    busy[a] += dt
                if len(names) == 1:
                    alone[names[0]] += dt
                for i, a in enumerate(names):
                    for b in names[i + 1:]:
                        pair[(a, b)] += dt
            last = t
            live[n] += s
    heavy = [k for k, _ in busy.most_common(top)]
    mat = {}
    for a in heavy:
        row = {"in_flight_ms": round(busy[a] / 1e6, 3), "alone_share": round(alone[a] / busy[a], 3) if busy[a] else 0.0, "beside": {}}
        for b in heavy:
            if a == b:
                continue
            t = pair.get((a, b), 0) + pair.get((b, a), 0)
            if busy[a] and t / busy[a] >= 0.05:
                row["beside"][b] = round(t / busy[a], 3)
        row["beside"] = dict(sorted(row["beside"].items(), key=lambda kv: -kv[1]))
        mat[a] = row
    return mat


def check_counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                calls[k] += 1
    out = {}
    for k, c in acc.items():
        e = {"dispatches": calls[k]}
        if c.get("SQ_BUSY_CU_CYCLES") and c.get("SQ_WAVE_CYCLES"):
            e["mean_waves_per_busy_cu"] = round(c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CU_CYCLES"], 2)
        if c.get("SQ_LEVEL_WAVES") and c.get("SQ_BUSY_CYCLES"):
            e["level_waves_per_busy_se_cycle"] = round(c["SQ_LEVEL_WAVES"] / c["SQ_BUSY_CYCLES"], 2)
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_BUSY_CU_CYCLES"):
            # SQ_BUSY_CU_CYCLES counts per CU (x4 sampling: one count = 4 cycles); GRBM_GUI_ACTIVE is chip cycles of the launch
            e["busy_cu_share"] = round(4.0 * c["SQ_BUSY_CU_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * CUS), 3)
        if c.get("SQ_WAVES"):
            e["waves_per_dispatch"] = round(c["SQ_WAVES"] / max(calls[k], 1), 1)
        if c.get("SQ_ACTIVE_INST_ANY") and c.get("SQ_WAVE_CYCLES"):
            e["active_share_of_wave_cycles"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        out[k] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "fiducials_amd", "lib", "libfid_amd.so"))
    ap.add_argument("--trace", nargs="*", default=[])
    ap.add_argument("--check", nargs="*", default=[])
    ap.add_argument("--label", default="")
    ap.add_argument("--launch-log", nargs="*", default=[], help="FID_LAUNCH_LOG files: 'kernel block dynamic_lds' per distinct launch shape")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        co = code_object(a.lib, td)
        notes = read_notes(co)
        tx = os.path.join(td, "text.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.text", co, tx])
        text_sha = hashlib.sha256(open(tx, "rb").read()).hexdigest()
    dm = demangle(list(notes))
    static = {}
    for mangled, meta in notes.items():
        static[short(dm.get(mangled, mangled))] = meta
    disp = read_trace(a.trace)
    dyn = collections.defaultdict(set)  # kernel base name -> {(block, dynamic LDS bytes)}
    for p in a.launch_log:
        for line in open(p):
            f = line.split()
            if len(f) == 3:
                dyn[f[0]].add((int(f[1]), int(f[2])))
    shapes = collections.defaultdict(lambda: collections.Counter())
    wgs = collections.defaultdict(list)
    for d in disp:
        shapes[d["name"]][(d["block"], d["lds"])] += 1
        wgs[(d["name"], d["block"], d["lds"])].append(d["wgs"])
    kernels = {}
    for name, meta in sorted(static.items()):
        e = {"vgpr": meta.get("vgpr_count", 0), "agpr": meta.get("agpr_count", 0), "sgpr": meta.get("sgpr_count", 0),
             "lds_static": meta.get("group_segment_fixed_size", 0), "scratch": meta.get("private_segment_fixed_size", 0),
             "max_block": meta.get("max_flat_workgroup_size", 0)}
        if meta.get("vgpr_spill_count") or meta.get("sgpr_spill_count"):
            e["spills"] = {"vgpr": meta.get("vgpr_spill_count", 0), "sgpr": meta.get("sgpr_spill_count", 0)}
        launches = []
        base = re.sub(r"<.*$", "", name)  # (the launch log names a kernel without its template arguments; "[g]" = group mode)
        if name in shapes:
            for (block, lds), n in shapes[name].most_common(6):
                # the trace's LDS_Block_Size is the STATIC group segment; the dynamic part comes from the library's own launch log
                dyns = sorted(d for b, d in dyn.get(base, ()) if b == block) or [0]
                for dl in dyns:
                    total = max(lds, e["lds_static"]) + dl
                    r = residency(e["vgpr"], e["agpr"], e["sgpr"], total, block)
                    w = sorted(wgs[(name, block, lds)])
                    r.update({"block": block, "lds_static": max(lds, e["lds_static"]), "lds_dynamic": dl, "lds_per_wg": total, "dispatches": n,
                              "wgs_per_launch_median": w[len(w) // 2], "wgs_per_launch_max": w[-1]})
                    if len(dyns) > 1:
                        r["note"] = "one of several dynamic sizes this kernel was launched with (dispatch counts are the kernel's, not the size's)"
                    r["rounds_to_drain"] = round(w[len(w) // 2] / max(r["wg_per_cu"] * CUS, 1), 2)
                    launches.append(r)
        else:
            r = residency(e["vgpr"], e["agpr"], e["sgpr"], e["lds_static"], e["max_block"] or 256)
            r.update({"block": e["max_block"] or 256, "lds_per_wg": e["lds_static"], "dispatches": 0, "note": "not in the trace: static LDS and launch bound only"})
            launches.append(r)
        e["launches"] = launches
        kernels[name] = e
    out = {"library_sha256": hashlib.sha256(open(a.lib, "rb").read()).hexdigest(), "device_text_sha256": text_sha, "label": a.label,
           "rules": {"vgpr": "min(8, 512 // ceil8(vgpr + agpr)) waves per SIMD", "lds": "163840 // ceil512(LDS per workgroup) workgroups per CU",
                     "waves": "32 per CU", "sgpr": "min(8, 800 // (ceil16(sgpr) + 16)) waves per SIMD",
                     "source": "/opt/skills/guides/MI355X_MICROARCH.md: Register files; Residency and cooperative launch"},
           "traces": [os.path.basename(p) for p in a.trace], "kernels": kernels}
    if disp:
        # (per trace file: the STag batch and the aruco bench are different runs, and the heavier one would crowd the other out of a joint top list)
        out["coresidency"] = {}
        for fn in sorted({d["file"] for d in disp}):
            out["coresidency"][fn] = coresidency([d for d in disp if d["file"] == fn], top=12)
    # what fits beside k_threshold_stream (the kernel the aruco walkers spend 40 - 65 % of their time beside): the cases behind DESIGN.md 4
    def shape(name):
        k = kernels.get(name)
        if not k:
            return None
        l = k["launches"][0]
        return {"vgpr": k["vgpr"], "agpr": k["agpr"], "sgpr": k["sgpr"], "lds": l.get("lds_per_wg", k["lds_static"]), "block": l["block"]}

    thr = shape("k_threshold_stream<3,4,13,3,false>")
    if thr and thr["lds"] > 0:
        out["beside_threshold"] = {}
        for n in (4, 3, 2, 1):
            row = {}
            for g in ("k_walk_full<2>", "k_seed_walk<true>", "k_seed_walk<false>", "k_seg_cycles<0u>", "k_seg_cycles<48u>", "k_probe_lut<6,0>", "k_approx", "k_seg_copy"):
                sh = shape(g)
                if sh:
                    row[g] = fits_beside(thr, n, sh)
            out["beside_threshold"][f"{n} threshold workgroups on the CU"] = row
    for p in a.check:
        out.setdefault("measured", {})[os.path.basename(os.path.dirname(p)) or os.path.basename(p)] = check_counters(p)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
