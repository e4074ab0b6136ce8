"""Randomised check of K5 (bf16 / fp16 GEMM with its five epilogues, two K segments, strided operands and outputs, the
narrow-tile and 8-wave routes) and of the fp32-output form against fp64 products computed by torch on the GPU.
Not part of the test suite; run on the GPU box:  python tools/gemm_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(cases):
    dt = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16])
    M = rnd.choice([1, 7, 255, 256, 257, rnd.randint(1, 3000), rnd.randint(3000, 20000)])
    N = 8 * rnd.choice([1, 2, 9, 16, 17, 32, 48, rnd.randint(1, 400), rnd.randint(400, 2700)])
    K = 64 * rnd.choice([1, 2, 3, 8, rnd.randint(1, 60), rnd.randint(60, 250)])
    epi = rnd.choice([0, 1, 2, 3, 4]) if dt == torch.bfloat16 else rnd.choice([0, 4])
    g = torch.Generator(device=dev).manual_seed(case)
    pad_a, pad_c = 8 * rnd.choice([0, 0, 1, 5]), 8 * rnd.choice([0, 0, 2]) + rnd.choice([0, 0, 4])
    abuf = (torch.randn(M, K + pad_a, generator=g, device=dev)).to(dt)
    a = abuf[:, :K]
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, generator=g, device=dev).to(dt) if rnd.random() < 0.8 else None
    cbuf = torch.full((M, N + pad_c), 3.0, dtype=dt, device=dev)
    c = cbuf[:, :N]
    kw = {}
    ref = a.double() @ w.double().T + (0 if b is None else b.double())
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    elif epi == 2:
        gf = 4 * rnd.randint(0, N // 4)
        kw["gelu_from"] = gf
        ref[:, gf:] = torch.nn.functional.gelu(ref[:, gf:], approximate="tanh")
    elif epi in (3, 4):
        kw["resid"] = torch.randn(M, N, generator=g, device=dev).to(dt)
        if epi == 3:
            kw["gate"] = torch.randn(N, generator=g, device=dev).to(dt)
            ref = kw["resid"].double() + kw["gate"].double() * ref
        else:
            ref = kw["resid"].double() + ref
    a2 = None
    if K >= 128 and rnd.random() < 0.3:                     # two K segments with one row stride
        k1 = 64 * rnd.randint(1, K // 64 - 1)
        two = torch.zeros(M, 2 * K + 64, dtype=dt, device=dev)
        two[:, :k1] = a[:, :k1]
        two[:, K + 64:K + 64 + (K - k1)] = a[:, k1:]
        a, a2 = two[:, :k1], two[:, K + 64:K + 64 + (K - k1)]
    out = ops.gemm(a, w, b, out=c, epilogue=epi, a2=a2, **kw)
    err = ((out.double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    worst = max(worst, err)
    ok = err < 6e-3 and bool(torch.isfinite(out.float()).all()) and bool((cbuf[:, N:] == 3.0).all())
    if ok and dt == torch.float16 and epi == 0 and b is None and a2 is None:
        f32 = torch.full((M, N + 4), 5.0, dtype=torch.float32, device=dev)
        ops.gemm_f16_f32out(a, w, f32[:, :N])
        e32 = ((f32[:, :N].double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        ok = e32 < 1e-5 and bool((f32[:, N:] == 5.0).all())
    if not ok:
        print(f"FAIL case {case}: dtype {dt} M={M} N={N} K={K} epi={epi} pads {pad_a},{pad_c} a2={a2 is not None} rel-L2 {err:.3e}")
        sys.exit(1)
print(f"{cases} cases passed; worst rel-L2 {worst:.2e}")
