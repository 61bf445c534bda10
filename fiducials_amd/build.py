"""Build libfid_amd.so (hand-written HIP kernels + C-ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The built library lands in fiducials_amd/lib/ (git-ignored, but it
travels to the GPU box with the working tree)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib", "libfid_amd.so")
SOURCES = ["fid_api.hip", "fid_kernels.hip", "fid_stag.hip", "fid_stag_route.hip", "fid_stag_lines.hip", "fid_stag_quads.hip",
           "fid_stag_pose.hip", "fid_jpeg.hip", "fid_png.hip", "fid_draw.hip", "fid_dict.hip", "fid_stag_batch.h", "fid_device.h", os.path.join("..", "..", "include", "fid_abi.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(SRC, s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        cmd = [HIPCC] + FLAGS + ["-o", LIB, os.path.join(SRC, "fid_api.hip"), "-lz"]  # zlib: fid_png.hip (host-side inflate)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=SRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
