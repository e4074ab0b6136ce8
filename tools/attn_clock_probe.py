"""Probe: one-round sweeps (every workgroup does the same 303 tiles) with 76 / 152 / 228 workgroups on 256 CUs.
If the time per sweep grows with the number of busy CUs, the kernel is limited by the chip (clock / power / memory),
not by its own instruction stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402
L = lib.load()
dev = "cuda:0"
S = 19385
g = torch.Generator(device=dev).manual_seed(0)
for H in (1, 2, 3, 24):
    d = H * 128
    q, k, v = (torch.randn(S, d, generator=g, device=dev).to(torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    for var in (0, 1):
        L.fp_debug_attn_variant(var)
        for _ in range(3):
            ops.attention(q, k, v, H, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.attention(q, k, v, H, out=out)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"H={H:2d} ({76 * H:4d} workgroups) variant {var}: {ms:.3f} ms  {4.0 * S * S * d / ms / 1e9:7.1f} TF/s", flush=True)
