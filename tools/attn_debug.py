"""Debug aid: error of the K1 variants against fp64 for a few shapes."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402
L = lib.load()
dev = "cuda:0"
for S, H in [(1, 1), (64, 1), (65, 1), (128, 1), (200, 2), (256, 1), (300, 1), (1000, 3)]:
    g = torch.Generator("cpu").manual_seed(S)
    q, k, v = (torch.randn(S, H * 128, generator=g).to(dev).bfloat16() for _ in range(3))
    qd, kd, vd = (t.double().view(S, H, 128).transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(1, 2) / math.sqrt(128.0), -1) @ vd).transpose(0, 1).reshape(S, H * 128)
    for var in (0, 1):
        L.fp_debug_attn_variant(var)
        o = ops.attention(q, k, v, H)
        torch.cuda.synchronize()
        err = (o.double() - ref).norm() / ref.norm()
        rows = ((o.double() - ref).norm(dim=1) / ref.norm(dim=1))
        bad = (rows > 2e-2).nonzero().flatten().tolist()
        print(f"S={S} H={H} variant {var}: rel err {err:.3e}  bad rows {len(bad)} {bad[:12]}", flush=True)
