#!/usr/bin/env python3
"""cfg 3 fed from host memory (bench.host_feed_result) on its own, with and without the chunked feed (run on the GPU box)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import bench
    from fiducials_amd.synth import K_DEFAULT
    B = int(os.environ.get("AB_BATCH", "256"))
    host = bench.make_frames(bench.shard_seeds(0, 1, B))
    print(json.dumps(bench.host_feed_result(0, host, K_DEFAULT.copy(), np.zeros(5))))
else:
    envs = [dict(kv.split("=", 1) for kv in a.split()) if a.strip() else {} for a in sys.argv[1:]] or [{}, {"FID_NO_FEED_OVERLAP": "1"}]
    for env in envs:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {"error": p.stderr[-300:]}
        print(env, {k: (v["value"], v["pcie_GBps"]) if isinstance(v, dict) else v[:40] for k, v in d.items()})
