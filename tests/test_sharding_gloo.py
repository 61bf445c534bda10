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
    # the bench's shard rule (cfg 4): stream s = rank owns seeds 10000*s + i
    seeds = [10000 * rank + i for i in range(4)]
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    got = [None] * world
    dist.all_gather_object(got, seeds)
    if rank == 0:
        print(json.dumps({"max_t": float(t.item()), "seeds": got}))
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
    assert r["max_t"] == 1.5
    flat = [x for s_ in r["seeds"] for x in s_]
    assert len(set(flat)) == len(flat) == 8  # disjoint shards, nothing exchanged but the clock
