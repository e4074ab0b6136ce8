"""Randomised check of K1 (both generations, bf16) and its fp8 variant's segment handling against fp64 attention:
random lengths, head counts, segment cuts (empty and one-row segments included), strided q|k|v views, output over q.
Not part of the test suite (takes a minute); run on the GPU box:  python tools/attn_fuzz.py [cases] [seed]"""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402

dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(cases):
    heads = rnd.choice([1, 2, 3, 5])
    kind = rnd.choice(["short", "tiles", "tiles", "long", "long", "huge"])
    S = {"short": rnd.randint(1, 200), "tiles": 64 * rnd.randint(1, 40) + rnd.choice([0, 0, 1, 63]),
         "long": rnd.randint(2000, 9000), "huge": rnd.randint(17000, 26000)}[kind]   # huge: the split last round
    nseg = rnd.choice([1, 1, 2, 3, 5])
    cuts = sorted(rnd.randint(0, S) for _ in range(nseg - 1))
    cu = [0] + cuts + [S]
    d = heads * 128
    g = torch.Generator().manual_seed(case)
    amp = rnd.choice([0.5, 1.0, 2.0])
    buf = (torch.randn(S, 3 * d + 64, generator=g) * amp).to(torch.bfloat16)
    bd = buf.to(dev)
    q, k, v = bd[:, :d], bd[:, d:2 * d], bd[:, 2 * d:3 * d]
    cu_t = torch.tensor(cu, dtype=torch.int32, device=dev) if nseg > 1 else None
    mx = max(b - a for a, b in zip(cu[:-1], cu[1:]))
    ref = torch.zeros(S, d, dtype=torch.float64, device=dev)        # fp64 on the GPU through torch (hipBLAS), not our kernels
    for a, b in zip(cu[:-1], cu[1:]):
        if b > a:
            qh, kh, vh = (bd[a:b, i * d:(i + 1) * d].double().reshape(b - a, heads, 128).transpose(0, 1) for i in range(3))
            p = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(128.0), dim=-1)
            ref[a:b] = (p @ vh).transpose(0, 1).reshape(b - a, d)
    for variant in (1, 2):
        lib.load().fp_debug_attn_variant(variant)
        out = ops.attention(q, k, v, heads, cu_seqlens=cu_t, max_seqlen=mx if nseg > 1 else None)
        err = ((out.double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        worst = max(worst, err)
        if not err < 1e-2 or not torch.isfinite(out.float()).all():
            print(f"FAIL case {case} variant {variant}: S={S} heads={heads} cu={cu} rel-L2 {err:.3e}")
            sys.exit(1)
    lib.load().fp_debug_attn_variant(2)
    o8 = ops.attention_fp8(q, k, v, heads, cu_seqlens=cu_t, max_seqlen=mx if nseg > 1 else None)
    e8 = ((o8.double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    if (amp <= 1.0 and not e8 < 0.2) or not torch.isfinite(o8.float()).all():
        print(f"FAIL fp8 case {case}: S={S} heads={heads} cu={cu} rel-L2 {e8:.3e}")
        sys.exit(1)
    # in place over q (the model's use)
    ops.attention(q, k, v, heads, out=q, cu_seqlens=cu_t, max_seqlen=mx if nseg > 1 else None)
    if not torch.equal(q, out):
        print(f"FAIL case {case}: output over q differs")
        sys.exit(1)
print(f"{cases} cases passed; worst bf16 rel-L2 {worst:.2e}")
