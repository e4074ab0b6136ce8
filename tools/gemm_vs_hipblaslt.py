"""Probe: the K5 GEMM beside torch's F.linear (hipBLASLt) at the DiT shapes, same process, interleaved."""
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


for M, N, K in [(19385, 21504, 3072), (19385, 3072, 15360), (18400, 9216, 3072), (18400, 12288, 3072), (18400, 3072, 12288),
                (18400, 3072, 3072), (8192, 8192, 8192), (16384, 8192, 3072), (16384, 4096, 3072)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    res = []
    L = lib.load()
    for rep in range(2):
        L.fp_debug_gemm_variant(0)
        t0 = timed(lambda: ops.gemm(a, w, b, out=out))
        o0 = out.clone()
        L.fp_debug_gemm_variant(1)
        t1 = timed(lambda: ops.gemm(a, w, b, out=out))
        same = torch.equal(o0, out)
        t2 = timed(lambda: torch.nn.functional.linear(a, w, b))
        res.append((t0, t1, t2))
    print(f"M={M} N={N} K={K} same={same}: " + " | ".join(f"8-wave {t0:6.3f} ms {fl / t0 / 1e9:5.0f} TF  4-wave {t1:6.3f} ms {fl / t1 / 1e9:5.0f} TF  hipBLASLt {t2:6.3f} ms {fl / t2 / 1e9:5.0f} TF" for t0, t1, t2 in res), flush=True)
