"""Price of the convolution's gather: the causal conv3d (implicit GEMM, A gathered) against a dense fp16 GEMM of the
same M, N, K.  usage: conv_vs_gemm_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (T, H, W, cin, cout) in ((4, 512, 512, 256, 256), (4, 512, 512, 128, 128), (4, 256, 256, 512, 512)):
    x = torch.randn(T, H, W, cin, generator=g, device=dev).half()
    w = (torch.randn(cout, 27 * cin, generator=g, device=dev) * (27 * cin) ** -0.5).half()
    b = torch.randn(cout, generator=g, device=dev).half()
    M, K = T * H * W, 27 * cin
    a = torch.randn(M, K, generator=g, device=dev).half()
    t_conv = timed(lambda: ops.conv3d_causal(x, w, b))
    t_gemm = timed(lambda: ops.gemm(a, w, b))
    fl = 2.0 * M * K * cout
    print(f"{T}x{H}x{W} {cin}->{cout}: conv {t_conv:.3f} ms ({fl / t_conv / 1e9:.0f} TF/s)  dense GEMM {t_gemm:.3f} ms "
          f"({fl / t_gemm / 1e9:.0f} TF/s)  ratio {t_conv / t_gemm:.3f}", flush=True)
    del a
