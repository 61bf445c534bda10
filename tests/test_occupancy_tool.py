"""tools/occupancy.py (round 6): the static half -- registers / LDS of every kernel from the SHIPPED code object's notes -- runs
without a GPU; the residency arithmetic (MI355X_MICROARCH.md: 512 registers per SIMD lane in granules of 8, 160 KB of LDS, 32 wave
slots per CU) is checked on the cases DESIGN.md quotes."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("occupancy", os.path.join(ROOT, "tools", "occupancy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_static_sheet_of_the_shipped_library():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "occupancy.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout)
    k = d["kernels"]
    for name in ("k_threshold_stream<3,4,13,3,false>", "k_seed_walk<false>", "k_walk_full<2>", "k_probe_lut<6,0>", "k_raw_to_gray", "k_probe_tables",
                 "k_stag_route_walk[g]", "k_stag_route_extract_big[g]", "k_stag_ccl_flatten[g]", "k_stag_refine[g]"):
        assert name in k, name
        assert 0 < k[name]["vgpr"] <= 512 and k[name]["launches"], name
    # the walkers' static LDS: 2 x 16 KB of windows + the 2 KB step table
    assert k["k_seed_walk<false>"]["lds_static"] == 34816 and k["k_walk_full<2>"]["lds_static"] == 34816
    # no kernel of the shipped library spills VECTOR registers or uses scratch memory (scalar registers spilled into vector lanes --
    # v_writelane / v_readlane, no memory -- occur in the largest kernels and are listed by the sheet)
    assert not [n for n, v in k.items() if v.get("spills", {}).get("vgpr")], [n for n, v in k.items() if v.get("spills", {}).get("vgpr")]
    # (scratch memory itself is used by a few kernels with dynamically indexed local arrays -- the simplex, the decoder, the rank sort --
    #  never by the streaming or the walking kernels)
    for name in ("k_threshold_stream<3,4,13,3,false>", "k_find_starts<true>", "k_seed_walk<false>", "k_walk_full<2>", "k_probe_lut<6,0>",
                 "k_stag_route_walk[g]", "k_stag_ccl_flatten[g]", "k_stag_smooth_grad[g]"):
        assert k[name]["scratch"] == 0, name
    assert len(d["device_text_sha256"]) == 64


def test_residency_arithmetic():
    occ = _tool()
    # k_threshold_stream: 124 VGPRs -> 128 allocated -> 4 waves per SIMD; 4-wave workgroups -> 4 per CU = the whole register file
    r = occ.residency(124, 0, 62, 31104, 256)
    assert r["waves_per_simd_by_vgpr"] == 4 and r["wg_per_cu"] == 4 and r["waves_per_cu"] == 16 and r["binding"] == "registers"
    # a 1 024-thread workgroup with 64 KB: two per CU (wave slots and LDS alike)
    r = occ.residency(50, 0, 27, 65536, 1024)
    assert r["wg_per_cu"] == 2
    thr = {"vgpr": 124, "agpr": 0, "sgpr": 62, "lds": 31104, "block": 256}
    walk = {"vgpr": 76, "agpr": 0, "sgpr": 80, "lds": 34816, "block": 256}
    seedw = {"vgpr": 80, "agpr": 0, "sgpr": 76, "lds": 34816, "block": 128}
    # beside FOUR threshold workgroups nothing fits (registers), beside three both walkers get 4 waves, beside two 8 against 4
    assert occ.fits_beside(thr, 4, walk)["waves"] == 0 and occ.fits_beside(thr, 4, seedw)["waves"] == 0
    assert occ.fits_beside(thr, 3, walk)["waves"] == 4 and occ.fits_beside(thr, 3, seedw)["waves"] == 4
    assert occ.fits_beside(thr, 2, walk)["waves"] == 8 and occ.fits_beside(thr, 2, seedw)["waves"] == 4
