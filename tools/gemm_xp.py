"""Times a few GEMM shapes with an experimental build of the library. usage: gemm_xp.py <lib.so> [tag]
(results of experimental builds are not checked: some leave out waits or loads on purpose)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib as fplib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    fplib.LIB_PATH = os.path.abspath(sys.argv[1])
from framepack_amd import ops  # noqa: E402

tag = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
if os.environ.get("FP_GEMM_VARIANT"):
    fplib.load().fp_debug_gemm_variant(int(os.environ["FP_GEMM_VARIANT"]))
    tag += " variant " + os.environ["FP_GEMM_VARIANT"]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
res = []
for M, N, K, epi in ((8192, 8192, 8192, 0), (19385, 3072, 15360, 3), (19385, 9216, 3072, 0), (19385, 21504, 3072, 2),
                     (18400, 3072, 12288, 3), (18400, 12288, 3072, 1), (18400, 3072, 3072, 3)):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = {}
    if epi == 3:
        kw = dict(resid=torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16), gate=b)
    if epi == 2:
        kw = dict(gelu_from=9216)
    if epi == 1:
        kw = {}
    for _ in range(3):
        ops.gemm(a, w, b, out=out, epilogue=epi, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, b, out=out, epilogue=epi, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res.append(f"{M}x{N}x{K}/e{epi} {ms:.3f} ms {2 * M * N * K / ms / 1e9:.0f} TF")
print(tag, " | ".join(res), flush=True)
