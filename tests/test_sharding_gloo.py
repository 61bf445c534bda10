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
