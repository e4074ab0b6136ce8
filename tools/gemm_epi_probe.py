"""Probe: cost of the K5 GEMM epilogues and of the K depth at the DiT's N = 3072 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for M, N, K in [(18400, 3072, 3072), (18432, 3072, 3072), (16384, 3072, 3072), (18400, 3072, 12288), (18400, 9216, 3072), (19385, 21504, 3072),
                (16384, 4096, 3072), (16384, 8192, 3072)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    gt = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    line = f"M={M} N={N} K={K} tiles={-(-M // 256) * -(-N // 256)} ({-(-M // 256) * -(-N // 256) / 256:.2f} rounds):"
    for epi, kw in [(0, {}), (1, {}), (3, dict(resid=r, gate=gt)), (4, dict(resid=r))]:
        t = timed(lambda: ops.gemm(a, w, b, out=out, epilogue=epi, **kw))
        line += f"  epi{epi} {t:6.3f} ms {fl / t / 1e9:5.0f} TF"
    print(line, flush=True)
