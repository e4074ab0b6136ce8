"""The single blocks' in-projection (M = 19 385, N = 21 504 = [k | v | q | mlp], K = 3072) as one launch - with the
tile-walk group sizes the diagnostics switch offers - against two launches ([k | v | q], then the MLP half with GELU).
Interleaved repetitions on one box.  usage: inproj_split_probe.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib as fplib, ops  # noqa: E402

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 19385
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(M, 3072, generator=g, device=dev)).to(torch.bfloat16)
W = (torch.randn(21504, 3072, generator=g, device=dev) * 3072 ** -0.5).to(torch.bfloat16)
b = torch.randn(21504, generator=g, device=dev).to(torch.bfloat16)
o1 = torch.empty(M, 21504, dtype=torch.bfloat16, device=dev)
o2 = torch.empty_like(o1)
L = fplib.load()


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fused(variant):
    def run():
        old = L.fp_debug_gemm_variant(variant)
        ops.gemm(x, W, b, out=o1, epilogue=2, gelu_from=9216)
        L.fp_debug_gemm_variant(old)
    return run


def split():
    ops.gemm(x, W[:9216], b[:9216], out=o2[:, :9216])
    ops.gemm(x, W[9216:], b[9216:], out=o2[:, 9216:], epilogue=1)


cases = {"one launch (default walk)": fused(1)}
for gm in (1, 2, 4, 8, 16):
    cases[f"one launch, groups of {gm}"] = fused(1 | gm << 4)
    cases[f"one launch, column-major groups of {gm}"] = fused(1 | 8 | gm << 4)
cases["two launches"] = split
res = {k: [] for k in cases}
for rep in range(3):
    for k, fn in cases.items():
        res[k].append(timed(fn))
for k, v in res.items():
    print(f"{k:45s} " + " ".join(f"{t:7.0f}" for t in v) + " us", flush=True)
split()
fused(1)()
torch.cuda.synchronize()
print("bit-identical:", torch.equal(o1, o2))
