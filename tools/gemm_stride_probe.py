"""Probe: does the row stride of the operands (L2 channel mapping) change the GEMM rate?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


L = lib.load()
for M, N, K in [(8192, 8192, 8192), (16384, 8192, 3072), (16384, 3072, 3072), (19385, 21504, 3072)]:
    fl = 2.0 * M * N * K
    line = f"M={M} N={N} K={K}:"
    for pad in (64,):
        a = torch.randn(M, K + pad, generator=g, device=dev).to(torch.bfloat16)[:, :K]
        w = (torch.randn(N, K + pad, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)[:, :K]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ts = []
        for v in (0, 1, 1):
            L.fp_debug_gemm_variant(v)
            ts.append(timed(lambda: ops.gemm(a, w, None, out=out)))
        t2 = timed(lambda: torch.nn.functional.linear(a, w))
        line += "  " + "/".join(f"{fl / t / 1e9:5.0f}" for t in ts) + f"/{fl / t2 / 1e9:5.0f}"
    print(line + "   (8-wave / 4-wave persistent x2 / hipBLASLt TF/s)", flush=True)
