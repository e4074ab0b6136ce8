"""Time of ONE tile per CU for a forced tile width (experimental libs built with -DFORCE_NT): M = 4096 (16 row tiles),
N = 16 column tiles of 32 NT columns, K = 8192.  usage: gemm_tile_probe.py <lib.so> <NT>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib as fplib  # noqa: E402

fplib.LIB_PATH = os.path.abspath(sys.argv[1])
from framepack_amd import ops  # noqa: E402

nt = int(sys.argv[2])
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 4096, 16 * 32 * nt, 8192
a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.gemm(a, w, b, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm(a, w, b, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
ref = (a[:64].float() @ w.float().T + b.float())
err = ((out[:64].float() - ref).norm() / ref.norm()).item()
print(f"NT={nt}: 256 tiles of 256x{32 * nt}, K={K}: {ms * 1e3:.1f} us per tile ({ms * 1e3 / (K // 64):.2f} us per K tile), "
      f"{2.0 * M * N * K / ms / 1e9:.0f} TF/s, err {err:.1e}", flush=True)
