"""N > 1 path: frames shard across ranks with no data-path collective; only the timing is max-reduced.
Covered here with world_size 2 on the gloo backend (CPU)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import bench  # the bench's own shard rule and job-rate reduction (no GPU needed for these two)
    seeds = bench.shard_seeds(rank, world, 4) + bench.shard_seeds(rank, world, 4, "stag")
    dist.barrier()
    rate, dt = bench.job_throughput(100, world, 0.5 + rank, dist, "cpu")
    got = [None] * world
    dist.all_gather_object(got, seeds)
    if rank == 0:
        print(json.dumps({"max_t": dt, "rate": rate, "seeds": got}))
    dist.destroy_process_group()
""") % ROOT


def test_world_size_2_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    wf = tmp_path / "worker.py"
    wf.write_text(WORKER)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(wf)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["max_t"] == 1.5 and abs(r["rate"] - 200 / 1.5) < 1e-9  # all ranks' units / the slowest rank's time
    flat = [x for s_ in r["seeds"] for x in s_]
    assert len(set(flat)) == len(flat) == 16  # disjoint shards (both workloads), nothing exchanged but the clock


def test_single_rank_keeps_the_cfg3_seeds():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_seeds(0, 1, 3) == [1000, 1001, 1002]
    assert bench.job_throughput(256, 1, 0.5) == (512.0, 0.5)


def _run_bench(argv, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_launcher_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no launcher around it must START two ranks (round 1 ran one process and multiplied):
    bench.py's own launcher, its rank plumbing and its clock reduction, on gloo with the GPU step replaced by a sleep."""
    import json
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "0"], {"FID_BENCH_DRYRUN": "cpu"})
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and len(r["ranks"]) == 2
    assert sorted(x["rank"] for x in r["ranks"]) == [0, 1]
    assert len({x["pid"] for x in r["ranks"]}) == 2  # two processes really ran
    # whole-job rate = all ranks' frames / the slowest rank's time: rank 1 sleeps twice as long as rank 0
    slow = min(x["fps"] for x in r["ranks"])
    assert r["value"] <= 2 * slow * 1.05 and r["value"] < sum(x["fps"] for x in r["ranks"])


def test_bench_refuses_a_rank_count_that_is_not_running():
    # launched by somebody else's launcher with a different world size
    out = _run_bench(["--gpus", "2"], {"FID_BENCH_DRYRUN": "cpu", "WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "WORLD_SIZE=3" in out.stderr
    # more GPUs asked for than visible (this container has none): no line, rc != 0
    out = _run_bench(["--gpus", "2", "--steps", "1"], {})
    assert out.returncode == 2 and "visible" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_eight_ranks_dry_run_seeds_budget_and_affinity():
    """The shape of the driver's 8-GPU run, on CPU (round-4 review, item 5d): `bench.py --gpus 8` starts EIGHT processes, every
    rank owns its own seed stream (cfg 4: stream s -> GPU s, nothing shared), the job's busy host threads stay inside what the
    host gives it, and every rank pins itself to the CPUs of its GPU's NUMA node (here: a made-up node map handed to the dry run,
    ranks 0-3 on node 0, 4-7 on node 1 where the machine has such nodes; ranks of one node get disjoint slices)."""
    import json
    out = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "0", "--feed", "jpeg"],
                     {"FID_BENCH_DRYRUN": "cpu", "FID_BENCH_DRYRUN_NODES": "0,0,0,0,1,1,1,1"}, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 8 and len(r["ranks"]) == 8 and r["feed"] == "jpeg"
    assert len({x["pid"] for x in r["ranks"]}) == 8
    seeds = [tuple(x["seeds"]) for x in sorted(r["ranks"], key=lambda x: x["rank"])]
    flat = [s_ for t in seeds for s_ in t]
    assert len(set(flat)) == len(flat) == 32 and seeds[3] == (30000, 30001, 30002, 30003)
    hb = r["host_budget"]
    assert hb["ranks"] == 8 and hb["threads_per_rank"] == max(1, hb["usable_cpus"] // 8)
    assert 1 <= hb["decoder_threads"] <= 3 and hb["decoder_contexts"] == hb["decoder_threads"] + 2
    assert hb["busy_host_threads_job"] <= max(hb["usable_cpus"], 8)  # (a host with fewer than 8 usable CPUs still runs one thread per rank)
    for x in r["ranks"]:
        assert x["pin"]["numa_node"] == (0 if x["rank"] < 4 else 1)
        if x["pin"]["cpus_pinned"]:
            assert x["affinity_cpus"] == x["pin"]["cpus_pinned"]
        # (round 6) a multi-GPU line explains itself: per rank the resident rate, the host-fed rate with the link's GB/s, the pin
        assert x["resident_fps"] > 0 and x["host_fed_fps"] > 0 and x["host_fed_pcie_GBps"] > 0
    assert r["host_fed"]["value"] > 0 and r["host_fed"]["unit"] == "frames/s"


def test_affinity_plan_is_disjoint_per_node():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    nodes = [0, 0, 1, 1, 1, -1]
    node_cpus = {0: list(range(0, 16)), 1: list(range(16, 40))}
    allowed = list(range(0, 36))  # (CPUs 36..39 are outside the process's mask)
    got = [bench.plan_affinity(r, 6, nodes, allowed, node_cpus)[0] for r in range(6)]
    assert got[0] == list(range(0, 8)) and got[1] == list(range(8, 16))
    assert got[2] == list(range(16, 22)) and got[3] == list(range(22, 28)) and got[4] == list(range(28, 34))
    assert got[5] is None  # no node reported: left alone
    assert bench._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    hb = bench.host_budget(8)
    assert hb["threads_per_rank"] * 8 <= max(hb["usable_cpus"], 8)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_with_real_kernels_on_one_gpu():
    """The N-rank path with real work (the lease is ONE GPU, so both ranks map to cuda:0 and the clock reduction runs on gloo:
    FID_BENCH_OVERSUBSCRIBE=1): `python bench.py --gpus 2` starts two processes through its own launcher, each loads torch's
    HIP runtime AND libfid_amd.so, owns its own stream of cfg 4 frames (seeds 10000 * rank + i) and finds 20 markers per frame.
    What is NOT executed here is the `nccl` branch of init_dist (RCCL refuses two ranks on one device)."""
    import json
    out = _run_bench(["--gpus", "2", "--batch", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                     {"FID_BENCH_OVERSUBSCRIBE": "1"}, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "oversubscribed" in r and r["n_gpus"] == 1 and len(r["ranks"]) == 2
    assert sorted(x["rank"] for x in r["ranks"]) == [0, 1] and len({x["pid"] for x in r["ranks"]}) == 2
    assert all(x["device"] == "cuda:0" and x["markers_per_frame_found"] == 20.0 for x in r["ranks"]), r["ranks"]
    assert r["config"]["frames_per_step"] == 64 and r["value"] > 0
    # whole-job rate = both ranks' frames / the slower rank's time
    assert r["value"] <= 2 * min(x["fps"] for x in r["ranks"]) * 1.05
    # (round 6) WITHOUT --no-extras -- what the driver's scaling run starts: after the timed region every rank also measures the
    # host-fed rate (pinned ring, fid_submit_batch), and the line carries both per rank with the link's GB/s and the pin
    out = _run_bench(["--gpus", "2", "--batch", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                     {"FID_BENCH_OVERSUBSCRIBE": "1"}, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert len(r["ranks"]) == 2 and r["host_fed"]["value"] > 0 and "extra" not in r
    for x in r["ranks"]:
        assert x["resident_fps"] > 0 and x["host_fed_fps"] > 0 and x["host_fed_pcie_GBps"] > 0 and "pin" in x
        assert x["markers_per_frame_found"] == 20.0
