#!/usr/bin/env python3
"""One-off source transformation (round 3): every `__global__ ... void k_stag_X(args) { body }` of the STag sources becomes
    __device__ __forceinline__ void k_stag_X_impl(args) { body }
    __global__ __launch_bounds__(N) void k_stag_X(args) { k_stag_X_impl(names); }          // the frame-at-a-time entry, as before
    struct k_stag_X_fn { kBounds = N; __device__ void operator()(args) const { k_stag_X_impl(names); } };   // for the batch trampoline
so that fid_stag_batch.h can run the same body for many frames in one launch.  Kept for the record; its output is committed."""
import re
import sys

HEAD = re.compile(r"^__global__ __launch_bounds__\((?P<b>[^\n]*?)\) void (?P<n>k_stag_\w+)\((?P<a>.*?)\)\n\{", re.S | re.M)


def split_args(a):
    out, depth, cur = [], 0, ""
    for ch in a:
        if ch in "(<[":
            depth += 1
        if ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def name_of(arg):
    return re.search(r"(\w+)\s*(\[[^\]]*\])?$", arg).group(1)


for path in sys.argv[1:]:
    s = open(path).read()
    pos, res = 0, ""
    n = 0
    while True:
        m = HEAD.search(s, pos)
        if not m:
            res += s[pos:]
            break
        end = s.index("\n}\n", m.end()) + 3  # the function's closing brace (column 0)
        args = " ".join(re.sub(r"/\*.*?\*/", "", m.group("a")).split())
        names = ", ".join(name_of(a) for a in split_args(args))
        b, nm = m.group("b"), m.group("n")
        res += s[pos:m.start()]
        res += f"__device__ __forceinline__ void {nm}_impl({m.group('a')})\n{{" + s[m.end():end]
        res += (f"__global__ __launch_bounds__({b}) void {nm}({args})\n{{\n    {nm}_impl({names});\n}}\n"
                f"struct {nm}_fn {{\n    static constexpr int kBounds = {b};\n"
                f"    __device__ __forceinline__ void operator()({args}) const {{ {nm}_impl({names}); }}\n}};\n")
        pos = end
        n += 1
    open(path, "w").write(res)
    print(path, n, "kernels")
