"""Micro-benchmark of the K1 attention kernel at the config-2 shape (S = 19385, 24 heads), random data.
usage: python tools/bench_attn.py [S] [heads] [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 19385
H = int(sys.argv[2]) if len(sys.argv) > 2 else 24
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
d = H * 128
if os.environ.get("LAYOUT", "fused") == "fused":
    buf = torch.randn(S, 7 * d, generator=g, device=dev).to(torch.bfloat16)   # single-block layout [k|v|q|mlp]
    k, v, q = buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:3 * d]
else:
    k, v, q = (torch.randn(S, d, generator=g, device=dev).to(torch.bfloat16) for _ in range(3))
out = torch.empty(S, d, dtype=torch.bfloat16, device=dev)
if os.environ.get("ATTN_VARIANT"):                     # 1: second-generation sweep, 2: third generation (default)
    from framepack_amd import lib as _fplib
    _fplib.load().fp_debug_attn_variant(int(os.environ["ATTN_VARIANT"]))
if os.environ.get("WS_SLOTS_PER_CU") == "1":             # round 3's first policy: one partial slot per CU (a partial last
    _full = ops._attn_workspace                         # round over half the CUs' worth then splits only an eighth)
    ops._attn_workspace = lambda device: _full(device)[:256 + (_full(device).numel() - 256) // 2]
attn = ops.attention_fp8 if os.environ.get("FP8") else ops.attention       # FP8=1: the e4m3 variant (quantisation included)
AB = [int(x) for x in os.environ.get("ATTN_AB", "").split(",") if x]      # e.g. ATTN_AB=18,2: alternate the variants run by run
for variant in range(int(os.environ.get("REPEAT", "2"))):
    if AB:
        from framepack_amd import lib as _fplib
        _fplib.load().fp_debug_attn_variant(AB[variant % len(AB)])
        print(f"variant {AB[variant % len(AB)]}:", end=" ")
    for _ in range(3):
        attn(q, k, v, H, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        attn(q, k, v, H, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    fl = 4.0 * S * S * H * 128
    # sampled-row check against fp64
    rows = torch.tensor([0, 255, 256, S // 2, S - 1], device=dev)
    qh, kh, vh = q[rows][:, :128].double(), k[:, :128].double(), v[:, :128].double()
    ref = torch.softmax(qh @ kh.T / math.sqrt(128.0), -1) @ vh
    err = ((out[rows][:, :128].double() - ref).norm() / ref.norm()).item()
    try:                                                    # probe builds (-DFP_V3_DIAG): in-kernel clock and cycles per tile
        import ctypes
        from framepack_amd import lib as _fplib
        dbg = (ctypes.c_ulonglong * 4)()
        if _fplib.load().fp_attn_dbg_read(dbg) == 0 and dbg[1] > 0:
            print(f"   in-kernel: {dbg[0] / dbg[2]:.0f} cycles per tile over {dbg[2]} tiles, clock {dbg[0] / dbg[1] * 0.1:.2f} GHz, slow-path tiles {dbg[3]}", flush=True)
    except AttributeError:
        pass
    print(f"run {variant}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF/s  ({fl / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)  err {err:.2e}", flush=True)
