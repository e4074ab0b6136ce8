"""Randomised check of K5 (bf16 / fp16 GEMM with its five epilogues, two K segments, strided operands and outputs, the
narrow-tile and 8-wave routes) and of the fp32-output form against fp64 products computed by torch on the GPU; the
GEMM with K2 in its epilogue against the two separate launches (same bits).
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
# K5 + K2 in one launch (fp_gemm_bf16_qknorm) against the two separate launches: bit-identical by construction
for case in range(max(cases // 5, 10)):
    heads = rnd.choice([1, 2, 3, 5])
    d = heads * 128
    M = rnd.choice([1, 255, 257, rnd.randint(1, 1500), rnd.randint(1500, 6000)])
    K = 64 * rnd.randint(1, 12)
    single = rnd.random() < 0.5
    if single:                                                 # [k | v | q | mlp ...], GELU on the tail
        N, qcol0, kcol0, epi, gf = 3 * d + 8 * rnd.randint(0, 80), 2 * d, 0, ops.EPI_GELU_FROM, 3 * d
    else:                                                      # [q | k | v]
        N, qcol0, kcol0, epi, gf = 3 * d, 0, d, ops.EPI_BIAS, 0
    rope_rows = rnd.choice([0, M, rnd.randint(0, M)])
    g = torch.Generator(device=dev).manual_seed(10_000 + case)
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16) if rnd.random() < 0.8 else None
    nw = [(torch.randn(128, generator=g, device=dev) * 0.2 + 1).to(torch.bfloat16) for _ in range(4)]
    txt = (nw[2], nw[3]) if rnd.random() < 0.5 else (None, None)
    ang = torch.rand(max(rope_rows, 1), 128, generator=g, device=dev) * 6.28
    rope = torch.cat([torch.cos(ang), torch.sin(ang)], dim=1).contiguous()
    pad = 8 * rnd.choice([0, 0, 3])
    want = torch.zeros(M, N + pad, dtype=torch.bfloat16, device=dev)[:, :N]
    ops.gemm(a, w, b, out=want, epilogue=epi, gelu_from=gf)
    ops.qk_rmsnorm_rope_(want[:, qcol0:qcol0 + d], want[:, kcol0:kcol0 + d], nw[0], nw[1], rope, rope_rows, heads,
                         wq_txt=txt[0], wk_txt=txt[1])
    got = torch.zeros(M, N + pad, dtype=torch.bfloat16, device=dev)[:, :N]
    ops.gemm_qknorm(a, w, b, got, nw[0], nw[1], rope, rope_rows, qcol0, kcol0, heads, epilogue=epi, gelu_from=gf,
                    wq_txt=txt[0], wk_txt=txt[1])
    if not torch.equal(got, want):
        print(f"FAIL qknorm case {case}: M={M} N={N} K={K} heads={heads} single={single} rope_rows={rope_rows}")
        sys.exit(1)
print(f"{cases} cases passed; worst rel-L2 {worst:.2e}")
