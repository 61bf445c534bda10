#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X fiducial detection hot path (BASELINE.json metric).

A "step" is one pass of the whole hot path (adaptive threshold x13 -> contour/quad extraction ->
identify -> subpix -> solvePnP) over one batch of synthetic 1920x1080 mono8 frames with 20
DICT_5X5_250 markers each (BASELINE cfg 3), the batch being resident in HBM when the timed region starts.
`value` = frames processed by all ranks / wall time (max over ranks), barrier + synchronize on both
sides.  With --gpus N every rank owns an independent stream of frames on its own GPU (cfg 4): no
data-path collective, scaling = "weak".

Also reported on the same JSON line:
  roofline      the dominant kernel's algorithmic bytes / its hipEvent-measured launch time vs 8 TB/s
  cpu_baseline  the CPU oracle (restatement of the reference's OpenCV path) timed on the host cores
                of this box on a bounded sample of the same frames (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("FID_PROFILE", "1")  # per-stage hipEvents on the context stream
# GPU_MAX_HW_QUEUES is NOT set here (rounds 1 - 3 defaulted it to 24): the line is measured with the HIP runtime's own default
# number of hardware queues, which is what a ROS node that links the library gets.  extra.hw_queues_24 re-runs the headline and
# the cfg 5 side result in child processes with GPU_MAX_HW_QUEUES=24 (the variable is read when the runtime starts).

import numpy as np  # noqa: E402

W, H, MARKERS = 1920, 1080, 20
FIDUCIAL_LEN = 0.14
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# FID_BENCH_DRYRUN=cpu: the launcher / rank plumbing only (gloo, no GPU, no detector, a sleeping stand-in for the step), so that
# the N-rank path of this very file is covered by the CPU test suite.  The line it prints carries "dryrun": true and no metric.
DRYRUN = os.environ.get("FID_BENCH_DRYRUN", "") == "cpu"
# FID_BENCH_OVERSUBSCRIBE=1: every rank maps to cuda:0 (the development lease is ONE GPU) and the clock reduction runs on gloo
# (RCCL refuses two ranks on one device).  It exists so that the N-rank path -- N processes, each with torch's HIP runtime AND
# libfid_amd.so and real kernels -- is executed by the GPU test suite; the line it prints says "oversubscribed" and n_gpus 1.
OVERSUB = os.environ.get("FID_BENCH_OVERSUBSCRIBE", "") == "1"

# algorithmic HBM bytes per frame and kernel (DESIGN.md "Roofline accounting"):
#   gray read once + 13 bit-packed masks written once + read once by the contour stage
N_SCALES = 13
MASK_BYTES = N_SCALES * W * H // 8
ALGO_BYTES = {
    "threshold": W * H + MASK_BYTES,  # K1: gray in, masks out
    "find_starts": MASK_BYTES,        # K2: masks in
    "walk_probe": MASK_BYTES,         # K3 probes: masks in (border pixels only; priced as one full read)
    "walk_full": MASK_BYTES,          # K3 survivor walk + link / chain / flatten: masks in (priced as one full read)
    "seed_walk": MASK_BYTES,          # K3 seed walk (own stream): masks in (border pixels only; priced as one full read)
    "approx": MASK_BYTES,             # K4: contour points of the accepted borders (priced as the masks they came from)
}
PIPELINE_BYTES = W * H + 2 * MASK_BYTES  # 8 812 800 B/frame
KERNEL_OF = {"threshold": "k_threshold_stream", "find_starts": "k_find_starts", "walk_probe": "k_probe",
             "walk_full": "k_walk_full<2>", "seed_walk": "k_seed_walk", "approx": "k_approx"}


def shard_seeds(rank: int, world: int, unique: int, workload: str = "aruco"):
    """The shard rule: which synthetic frames a rank owns.  Frames are independent units, every rank processes its own stream
    (BASELINE cfg 4: stream s -> GPU s), nothing is exchanged on the data path.  N = 1 keeps the cfg 3 seeds 1000 + i."""
    if workload == "stag":
        return [10000 * rank + 100 + i for i in range(unique)]
    return [1000 + i for i in range(unique)] if world == 1 else [10000 * rank + i for i in range(unique)]


def job_throughput(units_per_rank: int, world: int, dt_local: float, dist=None, device=None) -> tuple:
    """Whole-job rate: the units all ranks processed / the slowest rank's time (the only collective of the job)."""
    dt = dt_local
    if dist is not None:
        import torch

        tt = torch.tensor([dt_local], dtype=torch.float64, device="cpu" if (DRYRUN or OVERSUB) else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return units_per_rank * world / dt, dt


def pmc_traffic(stage, frames_per_launch):
    """HBM bytes per launch of a stage's kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    FETCH_SIZE / WRITE_SIZE per frame, collected in their own --pmc runs as MI355X_MICROARCH.md prescribes, corrected with
    the factors calibrated on known-byte-count kernels in this library's access patterns -- tools/gpu_pmc3.sh).  The file
    carries the hash of the library it was measured on: a file of another build is refused (None)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        import hashlib

        from fiducials_amd import _lib

        with open(path) as fh:
            doc = json.load(fh)
        if not _lib.profile_matches(doc):
            return None
        t = doc["per_frame_bytes"].get(stage)
        return None if t is None else int(t * frames_per_launch)
    except Exception:
        return None


def issue_roofline(fps_per_gpu):
    """The compute-side roofline: VALU wave-instructions per second against what the chip issues, MEASURED (tools/valu_calib.hip
    -> profiles/r04_valu_calib.json): gfx950 SIMDs issue a wave64 VALU instruction every 2 cycles only for a small class
    (v_add/sub_u32, and/or/xor/not, right shifts, v_mov, fp32 add/mul/fma: 900 - 1050 G/s on the chip) and every 4 cycles for
    the rest (mul24 / mad / dot2c / packed 16-bit / alignbit / bfe / left shifts / min / max / compares / selects / lane moves /
    DPP forms / f64: 570 - 590 G/s).  achieved = SQ_INSTS_VALU per frame (profiles/sq_cycles.json, rocprofv3 --pmc on a 64-frame
    sub-batch of THIS build, refused if the library hash differs) x frames/s; peak_measured = the harmonic mix of the two class
    peaks with every kernel's static class shares (profiles/isa_classes.json, disassembly of the shipped .so) weighted by its
    dynamic SQ_INSTS_VALU."""
    out = {"bound": "valu-issue", "unit": "G wave-instructions/s", "achieved": None, "peak_measured": None, "frac": None}
    try:
        import hashlib
        import statistics

        from fiducials_amd import _lib

        with open(os.path.join(ROOT, "profiles", "r04_valu_calib.json")) as fh:
            cal = json.load(fh)
        rates = {k.split()[0]: v["by_waves_per_simd"]["8"]["chip_ginstr_s"] for k, v in cal["kinds"].items()
                 if v["counts"] == "VALU" and k.startswith("v_") and " " not in k.strip()}
        fast = [r for r in rates.values() if r > 800]
        slow = [r for r in rates.values() if 300 < r <= 800]
        p2, p4 = statistics.median(fast), statistics.median(slow)
        out.update({"peak_2cycle_class": round(p2, 1), "peak_4cycle_class": round(p4, 1),
                    "calibration": "profiles/r04_valu_calib.json (tools/valu_calib.hip, 8 waves per SIMD, whole chip)"})
        with open(_lib.lib_path(), "rb") as fh:
            sha = hashlib.sha256(fh.read()).hexdigest()
        with open(os.path.join(ROOT, "profiles", "isa_classes.json")) as fh:
            cls = json.load(fh)
        with open(os.path.join(ROOT, "profiles", "sq_cycles.json")) as fh:
            sq = json.load(fh)
        if not _lib.profile_matches(cls) or not _lib.profile_matches(sq):
            out["note"] = "profiles/sq_cycles.json / isa_classes.json were measured on another build of the library: achieved unknown"
            return out
        by_mangled = {v["mangled"]: v for v in cls["kernels"].values()}
        valu = salu = t_fast = t_slow = 0.0
        for name, k in sq["kernels"].items():
            if not name.startswith("k_"):
                continue
            n = k.get("SQ_INSTS_VALU", 0.0)
            valu += n
            salu += k.get("SQ_INSTS_SALU", 0.0)
            base = name.split("<")[0]
            cand = [v for m, v in by_mangled.items() if base in m]
            # (round 5: every instruction weighted by 8 ^ loop depth -- tools/isa_classes.py fast_share_depth -- instead of counting a
            #  prologue like a loop body; files of before carry only the flat share)
            share = (sum(c["valu"] * (c.get("fast_share_depth") if c.get("fast_share_depth") is not None else c["fast"] / max(c["valu"], 1))
                         for c in cand) / max(sum(c["valu"] for c in cand), 1)) if cand else 0.0
            t_fast += n * share
            t_slow += n * (1.0 - share)
        peak = valu / (t_fast / p2 + t_slow / p4) if valu > 0 else None
        ach = valu * fps_per_gpu / 1e9
        out.update({"achieved": round(ach, 1), "peak_measured": round(peak, 1), "frac": round(ach / peak, 4),
                    "valu_wave_instr_per_frame": round(valu), "salu_wave_instr_per_frame": round(salu),
                    "static_2cycle_share": round(t_fast / valu, 3),
                    "class_mix": "per kernel: share of 2-cycle-class opcodes with every instruction weighted by 8 ^ (loop depth) in the shipped "
                                 "code object (tools/isa_classes.py), kernels weighted by their SQ_INSTS_VALU",
                    "counters": "profiles/sq_cycles.json (SQ_INSTS_VALU / SQ_INSTS_SALU per frame, one 64-frame sub-batch)"})
        # lane work under the wave-instructions (round 5): SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of one counter pass
        # = the share of a wave's 64 lanes that execute per VALU instruction, weighted over the pipeline's kernels by their
        # instruction counts; frac_lanes = frac x lane_util is the share of the chip's LANE-instruction rate that does work
        lu = sq.get("pipeline_lane_util")
        if lu is not None:
            out.update({"lane_util": round(lu, 4), "frac_lanes": round(ach / peak * lu, 4),
                        "lane_util_by_kernel": {n.split("(")[0][:40]: k["lane_util"] for n, k in sq["kernels"].items()
                                                if n.startswith("k_") and "lane_util" in k and k.get("SQ_INSTS_VALU", 0) > 0.01 * valu}})
    except Exception as e:  # noqa: BLE001
        out["note"] = f"unavailable: {e!r}"
    return out


def roofline_of(stage, stage_ms, launches, frames_per_launch):
    """The roofline object of one stage's kernel: algorithmic bytes per launch / the hipEvent-measured launch duration."""
    ms = stage_ms[stage] / launches  # average launch duration (hipEvents on the launching stream)
    achieved = ALGO_BYTES[stage] * frames_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"bound": "hbm", "kernel": KERNEL_OF.get(stage, "k_" + stage), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(stage, frames_per_launch),
            "algo_bytes_per_launch": int(ALGO_BYTES[stage] * frames_per_launch), "kernel_ms_per_launch": round(ms, 4),
            "launches_per_step": launches}


def cfg2_latency(local_rank, frame):
    """BASELINE cfg 2 (the node's imageCallback shape): one 1920x1080 frame per call on a max_batch 1 context, detect + pose,
    median over 30 calls after 5 warm-ups; from host memory (PCIe copy inside) and resident in HBM."""
    import torch

    from fiducials_amd.detector import ArucoDetector
    from fiducials_amd.synth import K_DEFAULT

    det = ArucoDetector("DICT_5X5_250", device=local_rank, max_width=W, max_height=H, max_batch=1, max_markers=64)
    dev = torch.from_numpy(frame).to(f"cuda:{local_rank}")
    torch.cuda.synchronize()
    out = {}
    for name, fn in (("host_frame_ms", lambda: det.detect_markers(frame)),
                     ("resident_frame_ms", lambda: det.detect_markers_device(dev.data_ptr(), 1, W, H))):
        ts = []
        for it in range(35):
            t = time.perf_counter()
            fn()
            det.pose_last(FIDUCIAL_LEN, K_DEFAULT, np.zeros(5), unpack=False)
            ts.append(time.perf_counter() - t)
        out[name] = round(float(np.median(ts[5:])) * 1e3, 3)
    det.close()
    return out



def usable_cpus():
    """(usable CPUs, hardware threads in the affinity mask, cgroup quota): what this process may really use -- the GPU boxes show
    256 hardware threads and carry a cgroup quota of 16 CPUs."""
    cores = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except Exception:  # noqa: BLE001
        pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    return max(1, int(min(cores, quota) if quota else cores)), cores, quota


def host_budget(world: int) -> dict:
    """What ONE rank may use of the host when `world` ranks share it (round-4 review: the compressed-stream figure used 3 decoder
    threads + 5 decoder contexts + 2 detector contexts per GPU, i.e. >= 32 busy host threads at 8 ranks on a cgroup quota of 16
    CPUs).  The job stays inside usable_cpus(): per rank usable // world host threads -- one drives the detector contexts, the rest
    (at most 3, at least 1) decode ahead; the frame-generation pool is sized the same way (make_frames)."""
    usable, cores, quota = usable_cpus()
    per_rank = max(1, usable // max(world, 1))
    dec = max(1, min(3, per_rank - 1))
    return {"usable_cpus": usable, "hardware_threads_in_mask": cores, "cgroup_cpu_quota": quota, "ranks": world,
            "threads_per_rank": per_rank, "decoder_threads": dec, "decoder_contexts": dec + 2, "detector_contexts": 2,
            "busy_host_threads_job": world * min(per_rank, dec + 1)}


def _parse_cpulist(text: str) -> list:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(n_devices: int, sysfs: str = "/sys") -> list:
    """NUMA node of every visible GPU (-1: the platform does not say): hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bdf>/numa_node."""
    nodes = []
    try:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        hip = None
    for d in range(n_devices):
        node = -1
        try:
            buf = ctypes.create_string_buffer(64)
            if hip is not None and hip.hipDeviceGetPCIBusId(buf, 64, d) == 0:
                with open(os.path.join(sysfs, "bus/pci/devices", buf.value.decode().lower(), "numa_node")) as fh:
                    node = int(fh.read().strip())
        except Exception:  # noqa: BLE001
            node = -1
        nodes.append(node)
    return nodes


def plan_affinity(local_rank: int, world: int, nodes: list, allowed: list, node_cpus: dict) -> tuple:
    """Which CPUs rank `local_rank` pins itself to: the CPUs of its GPU's NUMA node that the process may use, cut into equal
    slices among the ranks whose GPUs sit on the same node (their pinned staging buffers are then first-touched on that node and
    their threads do not migrate away from it).  Pure function of its arguments (tests/test_sharding_gloo.py).  Returns
    (cpus or None, note)."""
    if local_rank >= len(nodes) or nodes[local_rank] < 0:
        return None, "no NUMA node reported for this GPU: affinity left as it is"
    node = nodes[local_rank]
    cpus = sorted(set(node_cpus.get(node, [])) & set(allowed))
    peers = [r for r in range(min(world, len(nodes))) if nodes[r] == node]
    if not cpus or local_rank not in peers:
        return None, f"NUMA node {node} has no CPU this process may use: affinity left as it is"
    k, n = peers.index(local_rank), len(peers)
    per = len(cpus) // n
    if per < 1:
        return cpus, f"NUMA node {node}: {len(cpus)} CPUs shared by {n} ranks"
    return cpus[k * per:(k + 1) * per], f"NUMA node {node}: CPUs {cpus[k * per]}..{cpus[(k + 1) * per - 1]} ({per} of {len(cpus)}, {n} rank(s) on the node)"


def pin_rank(local_rank: int, world: int) -> dict:
    """Pins this rank (its threads inherit the mask; call before any pinned allocation or thread pool) to the NUMA node of its GPU."""
    info = {"numa_node": -1, "cpus_pinned": None, "note": ""}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if DRYRUN:
            nodes = [int(x) for x in os.environ.get("FID_BENCH_DRYRUN_NODES", "").split(",") if x.strip()] or [-1] * world
        else:
            import torch

            nodes = gpu_numa_nodes(torch.cuda.device_count())
        node_cpus = {}
        for nd in sorted(set(x for x in nodes if x >= 0)):
            try:
                with open(f"/sys/devices/system/node/node{nd}/cpulist") as fh:
                    node_cpus[nd] = _parse_cpulist(fh.read())
            except OSError:
                node_cpus[nd] = []
        cpus, note = plan_affinity(local_rank, world, nodes, allowed, node_cpus)
        info["numa_node"] = nodes[local_rank] if local_rank < len(nodes) else -1
        info["note"] = note
        if cpus and os.environ.get("FID_BENCH_NO_PIN", "0") != "1":
            os.sched_setaffinity(0, cpus)
            info["cpus_pinned"] = len(cpus)
    except Exception as e:  # noqa: BLE001
        info["note"] = f"not pinned: {e!r}"
    return info


def _gen_one(args):
    seed, dname = args
    from fiducials_amd.dictionary import get_predefined_dictionary
    from fiducials_amd.synth import make_frame

    return make_frame(get_predefined_dictionary(dname), seed, width=W, height=H, n_markers=MARKERS).image


def make_frames(seeds, dname="DICT_5X5_250"):
    """Synthetic frames, generated on the host cores with a process pool and cached."""
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fid_synth_{dname}_{W}x{H}_{MARKERS}")
    os.makedirs(cache, exist_ok=True)
    out = [None] * len(seeds)
    todo = []
    for i, s in enumerate(seeds):
        p = os.path.join(cache, f"{s}.npy")
        if os.path.exists(p):
            try:
                out[i] = np.load(p)
                continue
            except Exception:
                pass
        todo.append((i, s))
    if todo:
        import multiprocessing as mp

        world = max(1, int(os.environ.get("WORLD_SIZE", "1")))  # every rank generates its own stream: share the USABLE host cores
        nproc = max(1, min(len(todo), usable_cpus()[0] // world, 64))
        with mp.get_context("fork").Pool(nproc) as pool:
            imgs = pool.map(_gen_one, [(s, dname) for _, s in todo], chunksize=1)
        for (i, s), im in zip(todo, imgs):
            out[i] = im
            try:
                np.save(os.path.join(cache, f"{s}.npy"), im)
            except Exception:
                pass
    return np.stack(out)


_CPU_FRAMES = None  # inherited by the forked pool workers


def _cpu_one(i):
    import oracle
    from fiducials_amd.dictionary import get_predefined_dictionary

    t = time.perf_counter()
    ids, corners = oracle.detect(_CPU_FRAMES[0][i % len(_CPU_FRAMES[0])], get_predefined_dictionary("DICT_5X5_250"))
    for c in corners:
        oracle.solve_pnp_square(_CPU_FRAMES[1], _CPU_FRAMES[2], c, FIDUCIAL_LEN)
    return time.perf_counter() - t, len(ids)


def _cpu_chunk(idx):
    return [_cpu_one(i)[0] for i in idx]


def cpu_baseline(frames, K, D, budget_s=20.0):
    """The oracle (kind "port": CPU restatement of OpenCV 4.2 detectMarkers + solvePnP, the stand-in for the reference's CPU
    path since libopencv_aruco is not on the machine) on this box's host cores, as SURVEY.md 8d specifies: built here with
    -O3 -march=native, (i) one process: median per-frame time over >= 30 frames after 3 warm-ups, (ii) frame-parallel on one
    process per core, frames pre-split, wall clock over the whole pool after a warm-up round.  Bounded sample (~budget_s)."""
    global _CPU_FRAMES
    import multiprocessing as mp
    import tempfile

    import oracle

    flags = "gcc -O3 (shipped build)"
    try:
        oracle.use_library(oracle.build_native(os.environ.get("TMPDIR") or tempfile.gettempdir()))
        flags = "gcc -O3 -march=native, built on this host"
    except Exception as e:  # noqa: BLE001  (no compiler on the box: the shipped generic build is timed instead, and says so)
        print("cpu_baseline: native rebuild failed, timing the shipped build:", e, file=sys.stderr)
    _CPU_FRAMES = (frames, K, D)
    for i in range(3):
        _cpu_one(i)
    per = []
    t0 = time.perf_counter()
    while len(per) < 30 or (len(per) < 60 and time.perf_counter() - t0 < budget_s * 0.25):
        per.append(_cpu_one(len(per))[0])
    med = float(np.median(per))
    # what this process may really use: the affinity mask and the cgroup CPU quota (a pool of 64 processes on a quota of 16 CPUs
    # "scales" 12 x whatever the code does)
    usable, cores, quota = usable_cpus()
    # frame-parallel: one process per core, frames pre-split.  The port keeps a per-thread arena behind its malloc / free
    # (oracle/ora_arena.h: round 2's build gave its tens of MB per frame back to the kernel every time and 64 processes scaled
    # 13 x); the mallopt below is for what still goes through glibc.  The rates at nproc, nproc / 2 and nproc / 4 processes
    # are all reported, the best one is `value` (all hardware threads are not always the best choice on an SMT host).
    try:
        import ctypes

        libc = ctypes.CDLL(None)
        libc.mallopt(-3, 1 << 30)  # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)  # M_TRIM_THRESHOLD
    except Exception:  # noqa: BLE001
        pass
    best = None
    pool_rates = {}
    for T in sorted({max(1, usable), max(1, usable // 2), min(cores, 2 * usable)}, reverse=True):
        per_proc = max(2, min(12, int(budget_s * 0.2 / max(med * 1.5, 1e-3))))
        chunks = [[(p * per_proc + k) for k in range(per_proc)] for p in range(T)]
        with mp.get_context("fork").Pool(T) as pool:
            pool.map(_cpu_chunk, [[p] for p in range(T)], chunksize=1)  # warm-up round: every process has the library and a frame
            t = time.perf_counter()
            times = pool.map(_cpu_chunk, chunks, chunksize=1)
            wall = time.perf_counter() - t
        pool_rates[str(T)] = round(T * per_proc / wall, 2)
        if best is None or T * per_proc / wall > best[0] * best[1] / best[2]:
            best = (T, per_proc, wall, times)
    T, per_proc, wall, times = best
    nall = T * per_proc
    oracle.use_library(None)
    return {
        "value": round(nall / wall, 2), "unit": "frames/s", "cores": T, "kind": "port",
        "sample": f"{nall} frames of the bench batch ({per_proc} per process, pre-split) on {T} processes ({usable} usable CPUs: {cores} hardware threads in the affinity mask, cgroup quota {quota}), "
                  f"wall clock after a warm-up round; 1 process: median {med * 1e3:.1f} ms per frame over {len(per)} frames "
                  f"after 3 warm-ups = {1 / med:.2f} frames/s (oracle/*.c, {flags})",
        "value_1core": round(1 / med, 2),
        "pool_frames_per_s_by_processes": pool_rates,  # usable CPUs, half of them, twice as many (the best one is `value`)
        "host": {"hardware_threads": os.cpu_count(), "cgroup_cpu_quota": quota, "usable_cpus": usable},
        "pool_speedup_over_1core": round(nall / wall * med, 1),
        "ms_per_frame_1core_median": round(med * 1e3, 2),
        "ms_per_frame_in_pool_median": round(float(np.median([x for c in times for x in c])) * 1e3, 2),
    }


def host_feed_result(local_rank, host, K, D):
    """cfg 3 fed from HOST memory (what the node's imageCallback sees: frames arrive in host buffers): the same batch through
    fid_detect_batch + fid_pose_last.  The library sends the frames up sub-batch by sub-batch on a copy stream and starts a
    sub-batch when its frames have landed, so the PCIe copy of one runs under the kernels of another.  Never `value` (that is
    measured with the frames resident in HBM); this is the PCIe-inclusive rate, from pinned and from pageable memory."""
    import torch

    from fiducials_amd.detector import ArucoDetector

    B = len(host)
    det = ArucoDetector("DICT_5X5_250", device=local_rank, max_width=W, max_height=H, max_batch=B, max_markers=64, max_candidates=2048)
    out = {"workload": f"cfg3 from host memory: batch {B} x 1920x1080 mono8 ({B * W * H / 1e6:.0f} MB per step over PCIe), detect + pose"}
    pinned = torch.from_numpy(host).pin_memory()
    for name, arr in (("pinned", pinned.numpy()), ("pageable", host)):
        found = 0
        for _ in range(2):
            det.detect_markers_batch(arr, unpack=False)
            det.pose_last(FIDUCIAL_LEN, K, D, unpack=False)
        t = time.perf_counter()
        steps = 4
        for _ in range(steps):
            found += sum(det.detect_markers_batch(arr, unpack=False))
            det.pose_last(FIDUCIAL_LEN, K, D, unpack=False)
        dt = time.perf_counter() - t
        out[name] = {"value": round(B * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
                     "pcie_GBps": round(B * W * H * steps / dt / 1e9, 2), "markers_per_frame_found": round(found / (B * steps), 2)}
    det.close()
    # a STREAM of such batches: two contexts in turn (fid_submit_batch / fid_collect / fid_order_after) -- the copy of batch
    # k + 1 runs under the kernels of batch k, and the step is what the link takes (531 MB per batch)
    from fiducials_amd.pipeline import BatchPipeline

    with BatchPipeline("DICT_5X5_250", depth=2, fiducial_len=FIDUCIAL_LEN, K=K, D=D, device=local_rank, max_width=W, max_height=H,
                       max_batch=B, max_markers=64, max_candidates=2048) as pipe:
        bufs = [pinned.numpy(), torch.from_numpy(host.copy()).pin_memory().numpy()]  # (a capture ring of two batches)
        # (rounds 3 - 4 also reported a "pageable_stream": the same stream from pageable memory ran 12.5 / 14.6 / 21.0 k frames/s in
        #  three runs -- there the step is the HIP runtime's own staging of 531 MB per batch through its pinned bounce buffers on the
        #  calling thread, i.e. the host's memcpy bandwidth and whoever else uses the host, nothing this library schedules; a node
        #  that cares registers its capture ring (hipHostRegister), which is the pinned figure.  Dropped from the line in round 5.)
        # bayer_stream (round 6, ABI 7): the same ring read as bayer_rggb8 mosaics -- what a raw camera driver publishes.  The
        # demosaicing (cv_bridge::toCvCopy(msg, BGR8)) and BGR2GRAY are the device's first kernel, so a frame crosses the link as its
        # 2.07 MB of message bytes exactly like a mono8 frame (round 5 made a 6.2 MB BGR8 copy on one host thread first)
        for name, arrs, enc in (("pinned_stream", bufs, {}), ("bayer_stream", bufs, {"encoding": "bayer_rggb8"})):
            for k in range(3):
                pipe.push_host(arrs[k % 2], unpack=False, **enc)
            pipe.flush(unpack=False)
            steps, runs = 12, []
            for _ in range(3):  # (the link and the host's cores are shared with other tenants: the median of three runs)
                found = 0
                t = time.perf_counter()
                for k in range(steps):
                    done = pipe.push_host(arrs[k % 2], unpack=False, **enc)
                    found += sum(done[0]) if done else 0
                found += sum(sum(d[0]) for d in pipe.flush(unpack=False))
                runs.append((time.perf_counter() - t, found))
            dt, found = sorted(runs)[1]
            out[name] = {"value": round(B * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
                         "pcie_GBps": round(B * W * H * steps / dt / 1e9, 2), "markers_per_frame_found": round(found / (B * steps), 2),
                         "in_flight": 2, "runs_frames_per_s": [round(B * steps / r[0], 1) for r in runs]}
    return out


def jpeg_to_markers(local_rank, files, Q=1):
    """Compressed frames in HOST memory -> markers + poses (what the node does with `transport:=compressed`): fid_jpeg_decode
    leaves the gray images in HBM, fid_detect_device + fid_pose_last work on them in place.  Q > 1 cuts the batch into pieces with
    a decoder thread one piece ahead of the detector (two decoder contexts take turns) -- measured SLOWER than the plain
    sequence (256 frames: 13.4 k frames/s in one piece, 13.0 k in two, 11.5 k in four): both stages keep the whole chip busy, so
    there is nothing to hide one behind the other, and smaller pieces cost both their efficiency.  Q = 1 is what is reported."""
    import threading

    from fiducials_amd import jpeg as fj
    from fiducials_amd.detector import ArucoDetector
    from fiducials_amd.synth import K_DEFAULT

    B = len(files)
    per = (B + Q - 1) // Q
    parts = [files[k * per:(k + 1) * per] for k in range(Q) if files[k * per:(k + 1) * per]]
    decs = [fj.JpegDecoder(max_width=W, max_height=H, max_batch=per, device=local_rank) for _ in range(2)]
    det = ArucoDetector("DICT_5X5_250", device=local_rank, max_width=W, max_height=H, max_batch=per, max_markers=64, max_candidates=2048)
    D = np.zeros(5)

    def run():
        ready = [threading.Event() for _ in parts]
        freed = [threading.Event() for _ in parts]
        err = []

        def decode_side():
            try:
                for k, part in enumerate(parts):
                    if k >= 2:
                        freed[k - 2].wait()  # the decoder context of quarter k still holds quarter k - 2 until it is detected
                    decs[k % 2].decode(part, "mono8", to_host=False)
                    ready[k].set()
            except Exception as e:  # noqa: BLE001
                err.append(e)
                for ev in ready:
                    ev.set()

        th = threading.Thread(target=decode_side)
        th.start()
        found = 0
        for k, part in enumerate(parts):
            ready[k].wait()
            if err:
                break
            ptr, w, h, _, _ = decs[k % 2].device_ptr()
            found += sum(det.detect_markers_device(ptr, len(part), w, h, unpack=False))
            det.pose_last(FIDUCIAL_LEN, K_DEFAULT, D, unpack=False)
            freed[k].set()
        th.join()
        if err:
            raise err[0]
        return found

    run()
    t = time.perf_counter()
    steps, found = 3, 0
    for _ in range(steps):
        found += run()
    dt = time.perf_counter() - t
    for d in decs:
        d.close()
    det.close()
    return {"value": round(B * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "markers_per_frame_found": round(found / (B * steps), 2),
            "workload": f"{B} JPEG frames in host memory -> fid_jpeg_decode (device gray) -> fid_detect_device + fid_pose_last, {Q} pieces of {per}, "
                        "decoding one piece ahead of the detector"}


class JpegStream:
    """A STREAM of JPEG batches (the node with `transport:=compressed`, frames keep coming): `decoder_threads` host threads decode
    batches ahead of the detector, each call on a decoder context of its own (decoder_threads + 2 contexts in turn: a decoded batch
    stays in its context until its markers are out), the detector side keeps two batches in flight on two contexts
    (fid_submit_device / fid_collect / fid_order_after).  The entropy decoder's passes are latency-bound -- a batch decodes in 8 ms
    alone and in 13 ms beside the detector, whichever way it is scheduled -- so several of them side by side are what fills the
    chip: one decoder thread 18.0 k frames/s, two 19.3 k, three 20.0 k (tools/gpu_jpeg_stream_diag.py, round 4).  run(n) takes n
    batches through and returns the markers found; `bench.py --feed jpeg` times it as the step (decoder_threads from host_budget)."""

    def __init__(self, local_rank, files, decoder_threads=3):
        from fiducials_amd import jpeg as fj
        from fiducials_amd.detector import ArucoDetector

        self.files = files
        self.B = len(files)
        self.NT = max(1, int(decoder_threads))
        self.J = self.NT + 2
        self.decs = [fj.JpegDecoder(max_width=W, max_height=H, max_batch=self.B, device=local_rank) for _ in range(self.J)]
        self.dets = [ArucoDetector("DICT_5X5_250", device=local_rank, max_width=W, max_height=H, max_batch=self.B, max_markers=64,
                                   max_candidates=2048) for _ in range(2)]

    def run(self, n):
        import threading

        from fiducials_amd.synth import K_DEFAULT

        B, NT, J, decs, dets, files = self.B, self.NT, self.J, self.decs, self.dets, self.files
        D = np.zeros(5)
        ready = [threading.Event() for _ in range(n)]
        freed = [threading.Event() for _ in range(n)]
        err = []

        def decode_side(t):
            try:
                for k in range(t, n, NT):
                    if k >= J:
                        freed[k - J].wait()
                    decs[k % J].decode(files, "mono8", to_host=False)
                    ready[k].set()
            except Exception as e:  # noqa: BLE001
                err.append(e)
                for ev in ready:
                    ev.set()

        ths = [threading.Thread(target=decode_side, args=(t,)) for t in range(NT)]
        for th in ths:
            th.start()
        found = 0

        def collect(k):
            nonlocal found
            found += sum(dets[k % 2].collect(unpack=False))
            dets[k % 2].pose_last(FIDUCIAL_LEN, K_DEFAULT, D, unpack=False)
            freed[k].set()

        for k in range(n):
            ready[k].wait()
            if err:
                break
            if k >= 2:
                collect(k - 2)
            ptr, w, h, _, _ = decs[k % J].device_ptr()
            dets[k % 2].submit_device(ptr, B, w, h, after=dets[(k - 1) % 2])
        if not err:
            for k in range(max(n - 2, 0), n):
                collect(k)
        else:
            for ev in freed:
                ev.set()
        for th in ths:
            th.join()
        if err:
            raise err[0]
        return found

    def close(self):
        for d in self.decs:
            d.close()
        for d in self.dets:
            d.close()


def jpeg_stream_to_markers(local_rank, files, n_batches=12, decoder_threads=3):
    st = JpegStream(local_rank, files, decoder_threads)
    B, NT, J = st.B, st.NT, st.J
    st.run(NT + 2)
    t = time.perf_counter()
    found = st.run(n_batches)
    dt = time.perf_counter() - t
    st.close()
    return {"value": round(B * n_batches / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / n_batches * 1e3, 3),
            "markers_per_frame_found": round(found / (B * n_batches), 2), "decoder_threads": NT,
            "workload": f"a stream of {n_batches} batches of {B} JPEG frames in host memory -> fid_jpeg_decode (device gray) -> fid_submit_device / "
                        f"fid_collect + fid_pose_last: {NT} decoder threads ahead on {J} decoder contexts, two detector contexts in turn"}


def jpeg_files_of(frames, B):
    """The frames as compressed_image_transport sends them (libjpeg defaults: 4:2:0, quality 80); None without Pillow."""
    import io

    try:
        from PIL import Image
    except ImportError:
        return None
    files = []
    for k in range(B):
        b = io.BytesIO()
        Image.fromarray(np.stack([frames[k % len(frames)]] * 3, -1)).save(b, "JPEG", quality=80, subsampling=2)
        files.append(b.getvalue())
    return files


def jpeg_side_result(local_rank, frames):
    """The ingest in front of the hot path when the node runs with its launch default `transport:=compressed`
    (aruco_detect.launch:6): the bench frames as compressed_image_transport sends them (libjpeg defaults: 4:2:0, quality 80),
    decoded on the device to the gray image the detector takes (fid_jpeg_decode); beside it libjpeg-turbo itself (Pillow:
    the library cv::imdecode uses) on one host core.  Needs Pillow to write the files; skipped without it."""
    import io

    try:
        from PIL import Image
    except ImportError:
        return {"skipped": "Pillow is not importable here: no JPEG files to decode"}
    from fiducials_amd import jpeg as fj
    from fiducials_amd.detector import ArucoDetector

    B = min(256, len(frames))
    files = []
    for k in range(B):
        b = io.BytesIO()
        Image.fromarray(np.stack([frames[k]] * 3, -1)).save(b, "JPEG", quality=80, subsampling=2)
        files.append(b.getvalue())
    t = time.perf_counter()
    for f in files[:12]:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    cpu = (time.perf_counter() - t) / 12
    out = {"workload": f"{B} frames 1920x1080, JPEG 4:2:0 quality 80 ({sum(map(len, files)) // B} bytes each), decoded to the detector's gray image in HBM",
           "cpu_baseline": {"value": round(1.0 / cpu, 1), "unit": "frames/s", "cores": 1, "kind": "reference",
                            "sample": "12 frames through libjpeg-turbo (Pillow; the library behind cv::imdecode), decode + RGB"}}
    dec = fj.JpegDecoder(max_width=W, max_height=H, max_batch=B, device=local_rank)
    for _ in range(2):
        dec.decode(files, "mono8", to_host=False)
    t = time.perf_counter()
    for _ in range(4):
        dec.decode(files, "mono8", to_host=False)
    out["value"] = round(B * 4 / (time.perf_counter() - t), 1)
    out["unit"] = "frames/s"
    out["sync_rounds"] = dec.last_rounds()
    dec.close()
    # algorithmic HBM bytes per frame: coded bytes in, coefficients out and in (2 B each), planes out and in,
    # gray out -- 4:2:0: 1.5 samples per pixel
    algo = sum(map(len, files)) // B + int(W * H * 1.5) * (2 + 2 + 1 + 1) + W * H
    out["roofline"] = {"bound": "hbm", "kernel": "pipeline (entropy decoding: latency-bound passes over the coded bytes)",
                       "achieved": round(out["value"] * algo / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(out["value"] * algo / 1e9 / HBM_PEAK_GBS, 5), "algo_bytes_per_frame": algo, "traffic": None}
    try:
        out["jpeg_to_markers"] = jpeg_to_markers(local_rank, files)
    except Exception as e:  # noqa: BLE001
        out["jpeg_to_markers"] = {"error": repr(e)}
    try:
        out["jpeg_stream_to_markers"] = jpeg_stream_to_markers(local_rank, files, decoder_threads=host_budget(1)["decoder_threads"])
    except Exception as e:  # noqa: BLE001
        out["jpeg_stream_to_markers"] = {"error": repr(e)}
    one = fj.JpegDecoder(max_width=W, max_height=H, max_batch=1, device=local_rank)
    det = ArucoDetector("DICT_5X5_250", device=local_rank, max_width=W, max_height=H, max_batch=1, max_markers=64)
    ts, te = [], []
    for it in range(25):
        t = time.perf_counter()
        one.decode(files[it % B], "mono8", to_host=False)
        t1 = time.perf_counter()
        ptr, w, h, _, _ = one.device_ptr()
        det.detect_markers_device(ptr, 1, w, h)
        te.append(time.perf_counter() - t)
        ts.append(t1 - t)
    one.close()
    det.close()
    out["single_frame_decode_ms"] = round(float(np.median(ts[5:])) * 1e3, 3)
    out["single_frame_decode_and_detect_ms"] = round(float(np.median(te[5:])) * 1e3, 3)
    return out


STAG_HD, STAG_EC, STAG_MARKERS, STAG_UNIQUE = 21, 7, 12, 16


def make_stag_frames(seeds):
    """The cfg 5 frames: 1920x1080, every one of the 12 ids of library HD21 once per frame (round 2 drew 20 markers from the
    12 ids and checkDuplicate threw half of the rendered work away: the line said 20 markers and found 10)."""
    import multiprocessing as mp

    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fid_synth_stag_HD{STAG_HD}_{W}x{H}_{STAG_MARKERS}")
    os.makedirs(cache, exist_ok=True)
    out, todo = {}, []
    for sd in seeds:
        try:
            out[sd] = np.load(os.path.join(cache, f"{sd}.npy"))
        except Exception:  # noqa: BLE001
            todo.append(sd)
    if todo:
        with mp.get_context("fork").Pool(max(1, min(len(todo), (os.cpu_count() or 2), 32))) as pool:
            for sd, im in zip(todo, pool.map(_gen_stag_one, todo, chunksize=1)):
                out[sd] = im
                try:
                    np.save(os.path.join(cache, f"{sd}.npy"), im)
                except Exception:  # noqa: BLE001
                    pass
    return [out[sd] for sd in seeds]


def _gen_stag_one(sd):
    from fiducials_amd import stag as fstag, synth

    words = fstag.load_library(STAG_HD)
    ids = np.random.default_rng(sd).permutation(len(words) // 4)[:STAG_MARKERS]
    return synth.make_stag_frame(words, sd, W, H, STAG_MARKERS, ids=ids).image


def stag_pmc_traffic():
    """HBM bytes per frame of the STag pipeline from the committed PMC passes (profiles/stag_pmc_traffic.json, tools/stag_pmc.sh),
    refused (None) when it was measured on another build of the library."""
    try:
        import hashlib

        from fiducials_amd import _lib

        with open(os.path.join(ROOT, "profiles", "stag_pmc_traffic.json")) as fh:
            doc = json.load(fh)
        if not _lib.profile_matches(doc):
            return None
        return int(doc["pipeline_bytes_per_frame"])
    except Exception:  # noqa: BLE001
        return None


def stag_side_result(local_rank, args):
    """BASELINE cfg 5 inside the default line (so that the driver's run times it too): a short run of the stag_detect path
    (Stag::detectMarkers + 5-point pose, frames from host memory, one frame per call on concurrent contexts) and the
    REFERENCE's own Stag::detectMarkers on one host core next to it."""
    from fiducials_amd import stag as fstag, synth

    # (frames as a grid dimension: 256 frame slots = eight groups of 32 in lockstep, each group one stream and one host thread, every
    #  kernel launched once per group -- round 6: a launch carries frame 0's arguments and 32 bytes per further frame, so a group is no
    #  longer held to the 16 frames whose argument tuples fit 4 KB.  Rounds 3 - 5: 128 slots = eight groups of 16, 256 frames per step.)
    hd, ec, B, T = STAG_HD, STAG_EC, 1024, 256  # (r5 shape 256 / 128: 5.5 k frames/s with this library; 1 024 / 256: 7.3 k; 2 048 / 256: 7.6 k)
    frames = make_stag_frames(shard_seeds(0, 1, STAG_UNIQUE, "stag"))
    pool = fstag.StagPool(hd, ec, n_contexts=T, max_width=W, max_height=H, device=local_rank)
    batch = np.stack([frames[i % len(frames)] for i in range(B)])
    pool.detect_markers_batch(batch, synth.K_DEFAULT, None, 0.18)
    t = time.perf_counter()
    steps, found = 2, 0
    for _ in range(steps):
        m, _ = pool.detect_markers_batch(batch, synth.K_DEFAULT, None, 0.18)
        found += sum(len(x) for x in m)
    dt = time.perf_counter() - t
    # (a blocking call starts and drains the groups' pipelines once per call: twice the frames per call shows how much of the figure
    #  above that is -- a stream of frames would have neither)
    batch2 = np.concatenate([batch, batch])
    t2 = time.perf_counter()
    pool.detect_markers_batch(batch2, synth.K_DEFAULT, None, 0.18)
    fps_2048 = len(batch2) / (time.perf_counter() - t2)
    del batch2
    pool.close()  # (before the next context: 22 + 1 streams would touch the limit of 24 hardware queues)
    one = fstag.StagDetector(hd, ec, max_width=W, max_height=H, device=local_rank)
    ts = []
    for i in range(12):
        t = time.perf_counter()
        one.detect_markers(frames[i % len(frames)])
        ts.append(time.perf_counter() - t)
    queued, rerun = one.queue_stats()  # (frames enqueued ahead of their own counts / how many of them had to be run again)
    one.close()
    algo = 10 * W * H  # ~10 B/px for the EDPF streaming stages (20.7 MB per 1080p frame)
    fps5 = B * steps / dt
    res = {"value": round(fps5, 2), "unit": "frames/s", "ms_single_frame": round(float(np.median(ts[2:])) * 1e3, 3),
           "single_frame_queued_ahead": {"calls": len(ts), "queued": queued, "rerun_on_the_counted_road": rerun},
           "frames_per_s_at_2048_per_call": round(fps_2048, 2),
           "roofline": {"bound": "hbm", "kernel": "pipeline (latency-bound: edge routing, line fitting, simplex search)",
                        "achieved": round(fps5 * algo / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fps5 * algo / 1e9 / HBM_PEAK_GBS, 6),
                        "algo_bytes_per_frame": algo, "traffic": stag_pmc_traffic()},
           "workload": f"cfg5: {B} frames per step ({len(frames)} unique) through {T} frame slots = groups of 32 in lockstep (frames as a grid "
                       "dimension), 1920x1080 mono8 from host memory, "
                       f"{STAG_MARKERS} markers per frame = every id of library HD21 once, errorCorrection 7",
           "markers_per_frame_rendered": STAG_MARKERS,
           "markers_per_frame_found": round(found / (B * steps), 2),
           "markers_note": "found < rendered is PARITY, not a miss of this library: the reference's own Stag::detectMarkers (oracle/_ref) finds the "
                           "same markers on these frames (tests/test_gpu_stag.py::test_the_benchmarked_cfg5_frames_equal_the_reference demands len(M) == len(ref))"}
    if not args.no_cpu_baseline:
        from oracle import stag_ref

        if stag_ref.available():
            stag_ref.detect_markers(frames[0], hd, ec)
            t = time.perf_counter()
            k = 0
            while time.perf_counter() - t < 4.0:
                stag_ref.detect_markers(frames[k % len(frames)], hd, ec)
                k += 1
            res["cpu_baseline"] = {"value": round(k / (time.perf_counter() - t), 2), "unit": "frames/s", "cores": 1, "kind": "reference",
                                   "sample": f"{k} frames through the reference's own Stag::detectMarkers (oracle/_ref), one thread, ~4 s"}
    return res


def main_stag(args):
    """BASELINE cfg 5 (a parity configuration, not the headline metric): 1920x1080 frames with 20 STag HD21 markers through
    fid_stag_detect_markers + fid_stag_pose_last, one frame per call (the stag_detect node's shape; the frame comes from host
    memory, so the PCIe copy is inside the number).  cpu_baseline = the REFERENCE's own Stag::detectMarkers (oracle/_ref: its
    sources compiled in place) on one host core."""
    rank, local_rank, world = rank_env(args)
    n_gpus = world
    import torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = init_dist(world, local_rank)
    from fiducials_amd import stag as fstag, synth

    hd, ec, B = STAG_HD, STAG_EC, min(args.batch, 256)
    frames = make_stag_frames(shard_seeds(rank, world, min(B, STAG_UNIQUE), "stag"))
    # several contexts side by side (fid_stag_detect_markers_batch: one host thread + one HIP stream per context inside the
    # library): a frame's work is a chain of small kernels, several frames in flight fill the GPU
    T = max(1, min(args.streams, B))
    pool = fstag.StagPool(hd, ec, n_contexts=T, max_width=W, max_height=H, device=local_rank)
    K = synth.K_DEFAULT
    batch = np.stack([frames[i % len(frames)] for i in range(B)])

    def step():
        m, _ = pool.detect_markers_batch(batch, K, None, 0.18)
        return sum(len(x) for x in m)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    markers = 0
    for _ in range(args.steps):
        markers += step()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    barrier()
    fps, dt = job_throughput(B * args.steps, n_gpus, time.perf_counter() - t0, dist, f"cuda:{local_rank}")
    ranks = gather_ranks(dist, rank, f"cuda:{local_rank}", B * args.steps, dt_local)
    assert len(ranks) == n_gpus
    if rank == 0:
        algo = 10 * W * H  # ~10 B/px for the EDPF streaming stages
        out = {
            "metric": "frames/sec @1920x1080 12 STag HD21 markers (stag_detect path: detectMarkers + 5-point pose)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": f"synthetic ({len(frames)} unique frames per GPU, host memory)",
            "config": {"workload": f"cfg5: {B} frames per step through {T} frame slots (groups of 16 in lockstep, frames as a grid dimension), 1920x1080 mono8, 12 markers/frame (every HD21 id once), "
                                   "library HD21, errorCorrection 7, marker_size 0.18", "frames_per_step": B * n_gpus, "contexts_per_gpu": T,
                       "parallelism": f"frames sharded over {n_gpus} GPU(s), no collective",
                       "markers_per_frame_found": round(markers / max(B * args.steps, 1), 2)},
            "roofline": {"bound": "hbm", "kernel": "pipeline (latency-bound: edge routing, line fitting, simplex search)",
                         "achieved": round(fps / n_gpus * algo / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(fps / n_gpus * algo / 1e9 / HBM_PEAK_GBS, 6), "traffic": None},
            "ranks": ranks,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import stag_ref

            if stag_ref.available():
                stag_ref.detect_markers(frames[0], hd, ec)
                t = time.perf_counter()
                k = 0
                while time.perf_counter() - t < 10.0:
                    stag_ref.detect_markers(frames[k % len(frames)], hd, ec)
                    k += 1
                out["cpu_baseline"] = {"value": round(k / (time.perf_counter() - t), 2), "unit": "frames/s", "cores": 1, "kind": "reference",
                                       "sample": f"{k} frames through the reference's Stag::detectMarkers (oracle/_ref/libstag_ref.so: the "
                                                 "reference sources compiled in place, OpenCV calls restated), one thread (the reference keeps "
                                                 "global state, PoseRefiner.cpp:9), ~10 s"}
        print(json.dumps(out))
    pool.close()
    if dist is not None:
        dist.destroy_process_group()


def free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpus() -> int:
    if DRYRUN:
        return 1 << 20
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed launcher: start N ranks of this very script, one per GPU,
    exactly as the driver does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...`), and hand their exit code back.  Refuses (rc 2) when fewer than N GPUs are visible: a rank
    count that did not run is never reported."""
    have = visible_gpus()
    if have < n and not (OVERSUB and have >= 1):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, FID_BENCH_CHILD="1"))


def rank_env(args):
    """(rank, local_rank, world) from the launcher's environment; the world size must be the --gpus the line will report."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a rank count that is not running",
              file=sys.stderr)
        sys.exit(2)
    if OVERSUB:
        local_rank = 0
    return rank, local_rank, world


def init_dist(world, local_rank):
    """Process group for the clock reduction (the job's only collective): RCCL on the GPUs, gloo in the CPU dry run."""
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist_mod

    if DRYRUN or OVERSUB:
        dist_mod.init_process_group("gloo")
    else:
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return dist_mod


def gather_ranks(dist, rank, device, units, dt_local, markers=None, extra=None):
    """Per-rank evidence for the JSON line: (rank, pid, device, frames/s of that rank alone, markers per frame it found)."""
    mine = {"rank": rank, "pid": os.getpid(), "device": device, "fps": round(units / dt_local, 2)}
    if markers is not None:
        mine["markers_per_frame_found"] = round(markers / max(units, 1), 2)
    if extra:
        mine.update(extra)
    if dist is None:
        return [mine]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, mine)
    return got


def main_dryrun(args):
    rank, local_rank, world = rank_env(args)
    pin = pin_rank(local_rank, world)  # (before anything allocates or starts threads)
    dist = init_dist(world, local_rank)
    seeds = shard_seeds(rank, world, 4)
    B = 4
    budget = host_budget(world)

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    dt_local = time.perf_counter() - t0
    barrier()
    fps, dt = job_throughput(B * args.steps, world, dt_local, dist, "cpu")
    # (the host-fed leg of a multi-GPU run, as a stand-in: the same keys the real run reports per rank)
    host_leg = None
    if world > 1 or os.environ.get("FID_BENCH_HOST_LEG") == "1":
        barrier()
        th = time.perf_counter()
        for _ in range(2):
            time.sleep(0.005 * (1 + rank))
        dth = time.perf_counter() - th
        barrier()
        hfps, hdt = job_throughput(B * 2, world, time.perf_counter() - th, dist, "cpu")
        host_leg = {"fps": round(B * 2 / dth, 2), "pcie_GBps": round(B * 2 * W * H / dth / 1e9, 4), "job_fps": round(hfps, 2),
                    "job_ms_per_step": round(hdt / 2 * 1e3, 3), "steps": 2}
    ranks = gather_ranks(dist, rank, "cpu", B * args.steps, dt_local,
                         extra={"seeds": seeds, "pin": pin, "affinity_cpus": len(os.sched_getaffinity(0)),
                                "resident_fps": round(B * args.steps / dt_local, 2),
                                **({"host_fed_fps": host_leg["fps"], "host_fed_pcie_GBps": host_leg["pcie_GBps"]} if host_leg else {})})
    if rank == 0:
        line = {"dryrun": True, "n_gpus": world, "steps": args.steps, "value": round(fps, 2), "ms_per_step": round(dt / args.steps * 1e3, 3),
                "ranks": ranks, "seeds_rank0": seeds, "host_budget": budget, "feed": args.feed}
        if host_leg:
            line["host_fed"] = {"value": host_leg["job_fps"], "unit": "frames/s", "ms_per_step": host_leg["job_ms_per_step"], "steps": host_leg["steps"]}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def main_feed_jpeg(args, rank, local_rank, world, dist, host, pin, budget):
    """--feed jpeg: a step = one batch of JPEG files in host memory -> markers + poses on the host (JpegStream).  The third way a
    camera node can feed a GPU (resident / raw over PCIe / compressed); never the headline."""
    import torch

    B = len(host)
    files = jpeg_files_of(host, B)
    if files is None:
        print("bench.py: --feed jpeg needs Pillow to write the JPEG files", file=sys.stderr)
        sys.exit(2)
    st = JpegStream(local_rank, files, budget["decoder_threads"])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    st.run(max(args.warmup, st.NT + 2))
    barrier()
    t0 = time.perf_counter()
    found = st.run(args.steps)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    barrier()
    fps, dt = job_throughput(B * args.steps, world, time.perf_counter() - t0, dist, f"cuda:{local_rank}")
    ranks = gather_ranks(dist, rank, f"cuda:{local_rank}", B * args.steps, dt_local, found, extra={"pin": pin})
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec @1920x1080 20-marker (aruco detect + pose hot path), frames fed as JPEG files from host memory",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": 1 if OVERSUB else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": f"synthetic ({B} frames per GPU as JPEG 4:2:0 quality 80, {sum(map(len, files)) // B} bytes each)",
            "config": {"workload": f"cfg4 as a compressed stream: batches of {B} JPEG frames 1920x1080 in host memory -> fid_jpeg_decode (gray in HBM) "
                                   "-> fid_submit_device / fid_collect + fid_pose_last, DICT_5X5_250, aruco_detect node defaults",
                       "batch_per_gpu": B, "frames_per_step": B * world, "parallelism": f"frames sharded over {world} GPU(s), no collective",
                       "markers_per_frame_found": round(found / max(B * args.steps, 1), 2),
                       "feed": "jpeg: every step's frames are JPEG files in host memory, decoded on the device ahead of the detector, NOT the headline configuration",
                       "decoder_threads": st.NT, "decoder_contexts": st.J, "in_flight": 2},
            "ranks": ranks, "host_budget": budget}))
    st.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU (BASELINE cfg 3: 256)")
    ap.add_argument("--unique", type=int, default=0, help="unique synthetic frames per GPU (0 = batch); fewer are tiled")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("FID_BENCH_IN_FLIGHT", "2")),
                    help="contexts that take the steps in turn: step k + 1 is submitted before step k is collected (1 = one call after the other)")
    ap.add_argument("--feed", choices=["resident", "host", "jpeg"], default="resident",
                    help="resident (default, the metric): the batch lies in HBM when the timed region starts.  host: every step's frames "
                         "come from pinned HOST memory through fid_submit_batch (BASELINE cfg 4 as north_star words it: independent camera "
                         "streams, one PCIe link per GPU).  jpeg: every step's frames are JPEG files in host memory (the node's launch "
                         "default transport:=compressed), decoded on the device ahead of the detector by the rank's decoder threads "
                         "(host_budget).  Both say so in config.feed and are not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg 2 latency and cfg 5 (STag) side results")
    ap.add_argument("--stag-side-child", action="store_true", help=argparse.SUPPRESS)  # the cfg 5 side result, own process
    ap.add_argument("--streams", type=int, default=128, help="stag workload: frame slots per GPU (groups of 16 in lockstep)")
    ap.add_argument("--workload", choices=["aruco", "stag"], default="aruco",
                    help="aruco = the BASELINE.json metric (default); stag = BASELINE cfg 5, the stag_detect path")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # N ranks of this script, one per GPU; rank 0 prints the line
    if DRYRUN:
        return main_dryrun(args)
    if args.workload == "stag":
        return main_stag(args)
    if args.stag_side_child:
        import torch

        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        torch.cuda.init()  # (torch's HIP runtime first, then the library's: see tests/conftest.py)
        print(json.dumps(stag_side_result(lr, args)))
        return 0

    rank, local_rank, world = rank_env(args)
    n_gpus = world
    pin = pin_rank(local_rank, world)  # (before the frame pool, the pinned capture ring and the decoder threads exist)
    budget = host_budget(world)
    if visible_gpus() <= local_rank:
        print(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, {visible_gpus()} visible)", file=sys.stderr)
        sys.exit(2)

    B = args.batch
    unique = args.unique or B
    if usable_cpus()[0] < 8:
        unique = min(unique, 32)  # keep generation inside the time budget on small hosts
    if world > 1 and not args.unique:
        # N ranks share the host's usable cores (16 on the bench host) for frame generation: a fixed budget of ~512 frames for the
        # whole job (a 1080p frame takes ~1 s of one core), i.e. 64 unique frames per rank at 8 ranks, tiled to the batch
        unique = min(unique, max(16, 512 // world))
    unique = min(unique, B)
    # cfg 3: seeds 1000 + i ; cfg 4 stream s: 10000 * s + i
    seeds = shard_seeds(rank, world, unique)
    frames_u = make_frames(seeds)

    import torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = init_dist(world, local_rank)

    from fiducials_amd.detector import ArucoDetector
    from fiducials_amd.synth import K_DEFAULT

    K = K_DEFAULT.copy()
    D = np.zeros(5)
    reps = (B + unique - 1) // unique
    host = np.concatenate([frames_u] * reps)[:B]
    d_frames = torch.from_numpy(host).to(f"cuda:{local_rank}")
    torch.cuda.synchronize()
    feed_host = args.feed == "host"
    feed_jpeg = args.feed == "jpeg"
    # --feed host: a capture ring of two pinned batches per rank (a step's frames must stay put until its results are collected)
    h_ring = [torch.from_numpy(host).pin_memory().numpy(), torch.from_numpy(host.copy()).pin_memory().numpy()] if feed_host else None
    if feed_jpeg:
        return main_feed_jpeg(args, rank, local_rank, world, dist, host, pin, budget)
    # The steps go through `depth` contexts in turn (fiducials_amd/pipeline.py: fid_submit_device / fid_collect): step k + 1 is
    # enqueued before step k's results are fetched, so the latency-bound end of one batch runs under the front of the next.
    # Every step is a whole pass (gray .. pose, results on the host) over its own 256 frames; all K are finished inside the
    # timed region.  --in-flight 1 is one fid_detect_device + fid_pose_last after the other.
    from fiducials_amd.pipeline import BatchPipeline

    depth = max(1, min(args.in_flight, 8))
    pipe = BatchPipeline("DICT_5X5_250", depth=depth, fiducial_len=FIDUCIAL_LEN, K=K, D=D,
                         ordered=os.environ.get("FID_BENCH_UNORDERED", "0") != "1", device=local_rank, max_width=W,
                         max_height=H, max_batch=B, max_markers=64, max_candidates=2048,
                         max_contours=int(os.environ.get("FID_BENCH_MAX_CONTOURS", "0")))
    det = pipe.detectors[0]
    stage_acc = {}
    counted = [0, 0]  # markers, steps collected

    def take(done, acc=True):
        if done is None:
            return
        counted[0] += sum(done[0])
        counted[1] += 1
        if acc:
            for k, v in pipe.detectors[pipe.last_collected].stage_ms().items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def push(k):
        if feed_host:
            return pipe.push_host(h_ring[k % 2], unpack=False)
        return pipe.push(d_frames.data_ptr(), B, W, H, unpack=False)

    for k in range(max(args.warmup, depth if args.warmup else 0)):  # (every context sets up its streams on its first batch)
        push(k)
    pipe.flush(unpack=False)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        take(push(k))
    for done in pipe.flush(unpack=False):
        take(done)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    markers = counted[0]
    assert counted[1] == args.steps  # every step's results were fetched inside the timed region
    barrier()
    fps, dt = job_throughput(B * args.steps, n_gpus, time.perf_counter() - t0, dist, f"cuda:{local_rank}")
    # BASELINE cfg 4 is eight CAMERA streams: frames arrive in host memory, and on an 8-GPU node the honest per-GPU row is the
    # host-fed one (one PCIe link per GPU, the host's memory system shared by all ranks).  So a multi-GPU run measures it too, every
    # rank at the same time, OUTSIDE the timed region of `value`: the same batch from a pinned ring of two (fid_submit_batch), a few
    # steps.  Per rank: both rates, the link's GB/s, the NUMA pin.  (N = 1: extra.cfg3_from_host carries it; FID_BENCH_HOST_LEG=1 forces it.)
    host_leg = None
    if (n_gpus > 1 or os.environ.get("FID_BENCH_HOST_LEG") == "1") and not feed_host and not args.no_extras:
        ring = [torch.from_numpy(host).pin_memory().numpy(), torch.from_numpy(host.copy()).pin_memory().numpy()]
        hs = max(4, min(args.steps, 12))
        for k in range(3):
            pipe.push_host(ring[k % 2], unpack=False)
        pipe.flush(unpack=False)
        barrier()
        th = time.perf_counter()
        for k in range(hs):
            pipe.push_host(ring[k % 2], unpack=False)
        pipe.flush(unpack=False)
        torch.cuda.synchronize()
        dth_local = time.perf_counter() - th
        barrier()
        hfps, hdt = job_throughput(B * hs, n_gpus, time.perf_counter() - th, dist, f"cuda:{local_rank}")
        host_leg = {"fps": round(B * hs / dth_local, 2), "pcie_GBps": round(B * hs * W * H / dth_local / 1e9, 2), "steps": hs,
                    "job_fps": round(hfps, 2), "job_ms_per_step": round(hdt / hs * 1e3, 3)}
        del ring
    ranks = gather_ranks(dist, rank, f"cuda:{local_rank}", B * args.steps, dt_local, markers,
                         extra={"pin": pin, "resident_fps": round(B * args.steps / dt_local, 2),
                                **({"host_fed_fps": host_leg["fps"], "host_fed_pcie_GBps": host_leg["pcie_GBps"]} if host_leg else {})})
    assert len(ranks) == n_gpus  # every reported GPU ran its own rank

    if rank == 0:
        stage_ms = {k: v / max(args.steps, 1) for k, v in stage_acc.items()}
        launches = max(det.last_launches(), 1)  # sub-batches on separate streams: every kernel is launched this often per step
        frames_per_launch = B / launches
        # the dominant KERNEL: of the stages that are one kernel launched once per sub-batch (walk_probe is the two probe
        # levels, walk_full five kernels, approx two launches: their event brackets are not one kernel's duration).  The
        # rocprofv3 kernel stats of the same command (profiles/) name the same kernel.
        dom = max((k for k in ("threshold", "find_starts", "seed_walk") if k in stage_ms), key=lambda k: stage_ms[k])
        roof = roofline_of(dom, stage_ms, launches, frames_per_launch)
        # the one kernel of the path that streams (gray in, 13 bit-packed masks out): the HBM roofline proper
        roof["streaming_kernel"] = roofline_of("threshold", stage_ms, launches, frames_per_launch)
        roof["pipeline"] = {
            "algo_bytes_per_frame": PIPELINE_BYTES,
            "achieved": round(fps / n_gpus * PIPELINE_BYTES / 1e9, 2),
            "frac": round(fps / n_gpus * PIPELINE_BYTES / 1e9 / HBM_PEAK_GBS, 5),
        }
        # what actually binds this integer / bit path: instruction issue, priced against the MEASURED issue rate
        roof["issue"] = issue_roofline(fps / n_gpus)
        out = {
            "metric": "frames/sec @1920x1080 20-marker (aruco detect + pose hot path)" + (", frames fed from host memory" if feed_host else ""),
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": 1 if OVERSUB else n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": f"synthetic ({unique} unique frames per GPU" + (f", tiled to {B}" if unique < B else "") + ")",
            "config": {
                "workload": f"cfg3: batch {B} x 1920x1080 mono8 " + ("from pinned host memory (cfg 4's feed)" if feed_host else "resident in HBM") +
                            ", 20 markers/frame, DICT_5X5_250, "
                            "13 threshold scales, SUBPIX, ITERATIVE PnP (aruco_detect node defaults)",
                "batch_per_gpu": B,
                "frames_per_step": B * n_gpus,
                "parallelism": f"frames sharded over {n_gpus} GPU(s), no collective",
                "markers_per_frame_found": round(markers / max(B * args.steps, 1), 2),
                "feed": "host: every step's frames come from pinned host memory over PCIe (fid_submit_batch), NOT the headline configuration"
                        if feed_host else "resident: the batch lies in HBM when the timed region starts",
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
                "in_flight": depth,
                "in_flight_note": f"{depth} contexts take the steps in turn (fid_submit_device / fid_collect): ms_per_step is the "
                                  "timed region / steps, a throughput figure; one step alone is one_at_a_time.ms_per_step",
            },
            "roofline": roof,
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "ranks": ranks,
            "host_budget": budget,
        }
        if host_leg:
            out["host_fed"] = {"value": host_leg["job_fps"], "unit": "frames/s", "ms_per_step": host_leg["job_ms_per_step"], "steps": host_leg["steps"],
                               "per_rank": "ranks[].host_fed_fps / host_fed_pcie_GBps beside ranks[].resident_fps and ranks[].pin",
                               "note": "the same batch per rank from a pinned host ring (fid_submit_batch), all ranks at once, timed after and outside "
                                       "`value`'s region: cfg 4's eight camera streams are host-fed, so THIS is the per-GPU row a node of cameras sees; "
                                       "`value` is the resident rate the metric is defined on"}
        if OVERSUB:
            out["oversubscribed"] = f"{n_gpus} ranks on ONE GPU (FID_BENCH_OVERSUBSCRIBE=1, gloo clock reduction): a plumbing run, not a scaling point"
        if depth > 1 and not args.no_extras:
            # the same steps one call after the other on one context (fid_detect_device + fid_pose_last), outside the timed region
            k1 = max(3, min(args.steps, 10))
            for other in pipe.detectors[1:]:  # (their streams would still claim hardware queues beside this one's)
                other.close()
            pipe.detectors = pipe.detectors[:1]
            for _ in range(2):  # (the context makes its second sub-batch's streams on the first such call)
                det.detect_markers_device(d_frames.data_ptr(), B, W, H, unpack=False)
                det.pose_last(FIDUCIAL_LEN, K, D, unpack=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k1):
                det.detect_markers_device(d_frames.data_ptr(), B, W, H, unpack=False)
                det.pose_last(FIDUCIAL_LEN, K, D, unpack=False)
            d1 = (time.perf_counter() - t1) / k1
            out["one_at_a_time"] = {"frames_per_s": round(B / d1, 1), "ms_per_step": round(d1 * 1e3, 3), "steps": k1}
        if n_gpus == 1 and not args.no_extras:
            pipe.close()
            det = None
            out["extra"] = {"cfg2_single_frame": cfg2_latency(local_rank, frames_u[0])}
            try:
                out["extra"]["cfg3_from_host"] = host_feed_result(local_rank, host, K, D)
            except Exception as e:  # noqa: BLE001
                out["extra"]["cfg3_from_host"] = {"error": repr(e)}
            try:
                out["extra"]["jpeg_ingest"] = jpeg_side_result(local_rank, frames_u)
            except Exception as e:  # noqa: BLE001
                out["extra"]["jpeg_ingest"] = {"error": repr(e)}
            try:
                # in a process of its own: the contexts of the runs above keep their hardware queues after they are closed,
                # and 16 more streams on top oversubscribe the queues (measured: 310 frames/s in-process, 1300 alone)
                cmd = [sys.executable, os.path.abspath(__file__), "--stag-side-child", "--gpus", "1"]
                if args.no_cpu_baseline:
                    cmd.append("--no-cpu-baseline")
                env = dict(os.environ, LOCAL_RANK=str(local_rank))
                p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                if p.returncode != 0 or not lines:
                    raise RuntimeError(f"stag side run failed (rc {p.returncode}): {p.stderr[-400:]}")
                out["extra"]["cfg5_stag"] = json.loads(lines[-1])
            except Exception as e:  # noqa: BLE001
                out["extra"]["cfg5_stag"] = {"error": repr(e)}
            if "GPU_MAX_HW_QUEUES" not in os.environ:
                # the same two numbers with GPU_MAX_HW_QUEUES=24 (what rounds 1 - 3 set for themselves): child processes, because
                # the runtime reads the variable when it starts
                hq = {"GPU_MAX_HW_QUEUES": "24"}
                try:
                    env = dict(os.environ, LOCAL_RANK=str(local_rank), GPU_MAX_HW_QUEUES="24")
                    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "10", "--warmup", "3", "--batch", str(B),
                           "--unique", str(unique), "--no-extras", "--no-cpu-baseline"]
                    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                    if p.returncode != 0 or not lines:
                        raise RuntimeError(f"rc {p.returncode}: {p.stderr[-300:]}")
                    doc = json.loads(lines[-1])
                    hq["cfg3"] = {"value": doc["value"], "unit": "frames/s", "ms_per_step": doc["ms_per_step"], "steps": doc["steps"]}
                    cmd = [sys.executable, os.path.abspath(__file__), "--stag-side-child", "--gpus", "1", "--no-cpu-baseline"]
                    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                    if p.returncode != 0 or not lines:
                        raise RuntimeError(f"stag rc {p.returncode}: {p.stderr[-300:]}")
                    doc = json.loads(lines[-1])
                    hq["cfg5_stag"] = {"value": doc["value"], "unit": "frames/s", "ms_single_frame": doc.get("ms_single_frame")}
                except Exception as e:  # noqa: BLE001
                    hq["error"] = repr(e)
                out["extra"]["hw_queues_24"] = hq
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_u, K, D)
        print(json.dumps(out))
    if det is not None:
        pipe.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
