"""Micro-benchmark of the K5 GEMM kernel at the 13B DiT shapes, random data, with a sampled fp64 check.
usage: python tools/bench_gemm.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [  # (M, N, K, epilogue, name)
    (19385, 21504, 3072, ops.EPI_GELU_FROM, "single in-proj [k|v|q|mlp]"),
    (19385, 3072, 15360, ops.EPI_GATE_RESIDUAL, "single out-proj"),
    (18400, 9216, 3072, ops.EPI_BIAS, "double qkv (image)"),
    (18400, 12288, 3072, ops.EPI_GELU, "double ff1 (image)"),
    (18400, 3072, 12288, ops.EPI_GATE_RESIDUAL, "double ff2 (image)"),
    (18400, 3072, 3072, ops.EPI_GATE_RESIDUAL, "double attn out (image)"),
    (985, 9216, 3072, ops.EPI_BIAS, "double qkv (text)"),
    (4096, 4096, 4096, ops.EPI_BIAS, "4096^3"),
    (8192, 8192, 8192, ops.EPI_BIAS, "8192^3"),
]
AB = [int(x) for x in os.environ.get("GEMM_AB", "").split(",") if x]     # e.g. GEMM_AB=3,1: static vs dynamic tile schedule
from framepack_amd import lib as _fplib  # noqa: E402
for M, N, K, epi, name in [sh for sh in SHAPES for _ in range(max(1, 2 * len(AB)))]:
    if AB:
        AB.append(AB.pop(0))
        _fplib.load().fp_debug_gemm_variant(AB[-1])
        name = f"[variant {AB[-1]}] {name}"
    a = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = {}
    if epi == ops.EPI_GATE_RESIDUAL:
        kw = dict(resid=torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16),
                  gate=torch.randn(N, generator=g, device=dev).to(torch.bfloat16))
    if epi == ops.EPI_GELU_FROM:
        kw = dict(gelu_from=9216)
    for _ in range(2):
        ops.gemm(a, w, b, out=out, epilogue=epi, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, w, b, out=out, epilogue=epi, **kw)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    # sampled check (rows spread over the tile grid, all columns) against fp64
    rows = torch.tensor([0, 1, 17, 255, 256, M // 2, M - 2, M - 1], device=dev)
    ref = a[rows].double() @ w.double().T + b.double()
    if epi == ops.EPI_GELU:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    elif epi == ops.EPI_GELU_FROM:
        ref[:, 9216:] = torch.nn.functional.gelu(ref[:, 9216:], approximate="tanh")
    elif epi == ops.EPI_GATE_RESIDUAL:
        ref = kw["resid"][rows].double() + kw["gate"].double() * ref
    err = ((out[rows].double() - ref).norm() / ref.norm()).item()
    fl = 2.0 * M * N * K
    print(f"{name:44s} M={M:6d} N={N:6d} K={K:6d}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s "
          f"({fl / ms / 1e9 / 25:.1f}% of 2.5 PF)  rel err {err:.2e}", flush=True)
