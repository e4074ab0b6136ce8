"""One GEMM shape a few times (for rocprofv3 --pmc passes). usage: gemm_one.py M N K [epi] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
kw = {}
if epi >= 3:
    kw = dict(resid=torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16))
    if epi == 3:
        kw["gate"] = b
variants = [int(v) for v in os.environ.get("FP_GEMM_VARIANTS", "1").split(",")]
from framepack_amd import lib as fplib  # noqa: E402
for _ in range(iters):
    for v in variants:
        fplib.load().fp_debug_gemm_variant(v)
        ops.gemm(a, w, b, out=out, epilogue=epi, **kw)
    torch.nn.functional.linear(a, w, b)
torch.cuda.synchronize()
