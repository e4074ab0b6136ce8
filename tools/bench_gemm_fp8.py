"""Micro-benchmark of the K6 fp8 GEMM (and its row quantiser) at the 13B DiT shapes, beside the bf16 K5 GEMM.
usage: python tools/bench_gemm_fp8.py [iters]"""
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
    (8192, 8192, 8192, ops.EPI_BIAS, "8192^3"),
]


def timed(fn):
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


for M, N, K, epi, name in SHAPES:
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
    wq, wsc = ops.quantize_weight_fp8(w)
    aq = torch.empty(M, K, dtype=ops.FP8, device=dev)
    asc = torch.empty(M, dtype=torch.float32, device=dev)
    t_q = timed(lambda: ops.quantize_rows_fp8(a, out=aq, scale=asc))
    from framepack_amd import lib as fplib
    fplib.load().fp_debug_gemm_variant(0)          # the first-generation 8-wave pipeline
    t_8w = timed(lambda: ops.gemm_fp8(aq, asc, wq, wsc, b, out=out, epilogue=epi, **kw))
    o8w = out.clone()
    fplib.load().fp_debug_gemm_variant(int(os.environ.get("FP_GEMM_VARIANT", "1")))
    t_8 = timed(lambda: ops.gemm_fp8(aq, asc, wq, wsc, b, out=out, epilogue=epi, **kw))
    o8 = out.clone()
    t_16 = timed(lambda: ops.gemm(a, w, b, out=out, epilogue=epi, **kw))
    fl = 2.0 * M * N * K
    d = ((o8.float() - out.float()).norm() / out.float().norm()).item()
    print(f"{name:30s} M={M:6d} N={N:6d} K={K:6d}  bf16 {t_16:7.3f} ms {fl / t_16 / 1e9:7.0f} TF/s | "
          f"fp8 {t_8:7.3f} ms {fl / t_8 / 1e9:7.0f} TF/s (8-wave kernel {t_8w:7.3f} ms, max diff {(o8.float() - o8w.float()).abs().max().item():.3g}) | quantise {t_q:6.3f} ms "
          f"{(M * K * 3) / t_q / 1e6:6.0f} GB/s | fp8 vs bf16 rel-L2 {d:.4f}", flush=True)
