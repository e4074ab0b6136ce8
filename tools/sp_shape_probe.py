"""Per-rank compute of one DiT step under Ulysses sequence parallelism, emulated on ONE GPU: the GEMMs at the rank's
row count, the attention at the rank's head count, and the two copy passes the exchange still needs (packing q|k|v
into the per-peer send layout, unpacking the returned heads into the projection's input; the attention output itself
is written straight into the inverse exchange's send buffer).  The all-to-all itself cannot be measured on one GPU.
usage: sp_shape_probe.py [P ...]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402

dev = "cuda:0"
S, d = 19385, 3072
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g, device=dev) * scale).to(torch.bfloat16)


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


W = {n: rnd(n, k, scale=k ** -0.5) for n, k in ((21504, 3072), (9216, 3072), (12288, 3072))}
W[(3072, 15360)] = rnd(3072, 15360, scale=15360 ** -0.5)
W[(3072, 12288)] = rnd(3072, 12288, scale=12288 ** -0.5)
W[(3072, 3072)] = rnd(3072, 3072, scale=3072 ** -0.5)
bias = {n: rnd(n) for n in (21504, 9216, 12288, 3072)}
base = None
for P in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n = math.ceil(S / P)
    x3, x15, x12 = rnd(n, 3072), rnd(n, 15360), rnd(n, 12288)
    res = rnd(n, 3072)
    o21, o9, o12, o3 = (torch.empty(n, k, dtype=torch.bfloat16, device=dev) for k in (21504, 9216, 12288, 3072))
    t = {}
    t["in-proj x40"] = 40 * timed(lambda: ops.gemm(x3, W[21504], bias[21504], out=o21, epilogue=2, gelu_from=9216))
    t["out-proj x40"] = 40 * timed(lambda: ops.gemm(x15, W[(3072, 15360)], bias[3072], out=o3, epilogue=3, resid=res, gate=bias[3072]))
    t["qkv x20"] = 20 * timed(lambda: ops.gemm(x3, W[9216], bias[9216], out=o9))
    t["attn-out x20"] = 20 * timed(lambda: ops.gemm(x3, W[(3072, 3072)], bias[3072], out=o3, epilogue=3, resid=res, gate=bias[3072]))
    t["ff1 x20"] = 20 * timed(lambda: ops.gemm(x3, W[12288], bias[12288], out=o12, epilogue=1))
    t["ff2 x20"] = 20 * timed(lambda: ops.gemm(x12, W[(3072, 12288)], bias[3072], out=o3, epilogue=3, resid=res, gate=bias[3072]))
    h = 24 // P
    q, k, v = rnd(S, h * 128), rnd(S, h * 128), rnd(S, h * 128)
    t["attention x60"] = 60 * timed(lambda: ops.attention(q, k, v, h), n=4)
    ql, kl, vl = o9[:, :3072], o9[:, 3072:6144], o9[:, 6144:]                # the q|k|v GEMM's output, row stride 9216
    wn = rnd(128)
    rope = torch.rand(n, 256, device=dev)
    info = ""
    if P == 1:
        t["K2 x60"] = 60 * timed(lambda: ops.qk_rmsnorm_rope_(ql, kl, wn, wn, rope, n, 24))
    else:
        from framepack_amd.ulysses import pick_chunks
        ch = pick_chunks(S, h)
        hwc = h // ch * 128
        t[f"K2 into the send buffer x60 ({ch} chunks)"] = 60 * timed(
            lambda: ops.qk_rmsnorm_rope_pack(ql, kl, vl, wn, wn, rope, n, 24, P, ch))
        recv = rnd(P, n, hwc)
        t["unpack x60"] = 60 * timed(lambda: [o3.unflatten(1, (P, ch, hwc))[:, :, c].copy_(recv.permute(1, 0, 2))
                                              for c in range(ch)])
        old_way = 60 * (timed(lambda: ops.qk_rmsnorm_rope_(ql, kl, wn, wn, rope, n, 24)) + timed(lambda: torch.stack(
            [x.reshape(n, P, ch, hwc) for x in (ql, kl, vl)], dim=3).permute(2, 1, 0, 3, 4).contiguous()))
        info = f" (K2 in place + a packing pass, replaced: {old_way:.1f})"
        # the in-projection as two launches ([k | v | q], then the MLP half under the forward exchange: model._single_block)
        qkv_ms = timed(lambda: ops.gemm(x3, W[21504][:9216], bias[21504][:9216], out=o21[:, :9216]))
        mlp_ms = timed(lambda: ops.gemm(x3, W[21504][9216:], bias[21504][9216:], out=o21[:, 9216:], epilogue=1))
        sent = n * 9216 * 2 * (P - 1) / P / 1e6
        info += (f"; in-proj split x40: q|k|v {40 * qkv_ms:.1f} + mlp {40 * mlp_ms:.1f} (one launch: {t['in-proj x40']:.1f}); "
                 f"the mlp launch ({mlp_ms * 1e3:.0f} us) covers the forward exchange of {sent:.1f} MB per rank")
    tot = sum(t.values())
    base = base or tot * P
    print(f"P={P} rows/rank {n} heads/rank {h}: " + " | ".join(f"{k} {v:.1f}" for k, v in t.items()) +
          f" | total {tot:.1f} ms = {base / P / tot * 100:.0f} % of ideal{info}", flush=True)
