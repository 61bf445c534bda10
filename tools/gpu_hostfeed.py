"""Host-fed batch rates by the kind of host memory: pageable, pinned by torch (its own HIP runtime instance), pinned by the runtime
the library uses (hipHostMalloc from /opt/rocm's libamdhip64 through ctypes).  Run on the GPU box: python tools/gpu_hostfeed.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch

    from fiducials_amd.pipeline import BatchPipeline
    from fiducials_amd.synth import K_DEFAULT

    torch.cuda.init()
    B = 256
    host = bench.make_frames(bench.shard_seeds(0, 1, B))
    hip = C.CDLL("libamdhip64.so.7")  # the instance libfid_amd.so is linked against
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]

    def lib_pinned(src):
        p = C.c_void_p()
        rc = hip.hipHostMalloc(C.byref(p), src.nbytes, 0)
        assert rc == 0, rc
        a = np.ctypeslib.as_array((C.c_uint8 * src.nbytes).from_address(p.value)).reshape(src.shape)
        a[...] = src
        return a

    kinds = {
        "pageable": lambda: [host, host.copy()],
        "torch_pinned": lambda: [torch.from_numpy(host).pin_memory().numpy(), torch.from_numpy(host.copy()).pin_memory().numpy()],
        "lib_pinned": lambda: [lib_pinned(host), lib_pinned(host)],
    }
    D = np.zeros(5)
    with BatchPipeline("DICT_5X5_250", depth=2, fiducial_len=bench.FIDUCIAL_LEN, K=K_DEFAULT, D=D, device=0, max_width=bench.W,
                       max_height=bench.H, max_batch=B, max_markers=64, max_candidates=2048) as pipe:
        for name, make in kinds.items():
            arrs = make()
            for k in range(3):
                pipe.push_host(arrs[k % 2], unpack=False)
            pipe.flush(unpack=False)
            runs = []
            for _ in range(3):
                t = time.perf_counter()
                for k in range(12):
                    pipe.push_host(arrs[k % 2], unpack=False)
                pipe.flush(unpack=False)
                runs.append(round(B * 12 / (time.perf_counter() - t), 1))
            det = pipe.detectors[0]
            t = time.perf_counter()
            for k in range(4):
                det.detect_markers_batch(arrs[k % 2], unpack=False)
                det.pose_last(bench.FIDUCIAL_LEN, K_DEFAULT, D, unpack=False)
            one = round(B * 4 / (time.perf_counter() - t), 1)
            print(json.dumps({"memory": name, "stream_frames_per_s": runs, "one_call_after_the_other": one}), flush=True)


if __name__ == "__main__":
    main()
