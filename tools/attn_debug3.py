import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402
L = lib.load()
dev = "cuda:0"
S, H = 64, 1
g = torch.Generator("cpu").manual_seed(S)
q, k, v = (torch.randn(S, H * 128, generator=g).to(dev).bfloat16() for _ in range(3))
sc = (q.double() @ k.double().T) / math.sqrt(128.0)
P = torch.softmax(sc, -1)
ref = P @ v.double()
L.fp_debug_attn_variant(1)
o = ops.attention(q, k, v, H).double()
torch.cuda.synchronize()
for r in [0, 1, 2, 3, 4, 8, 9]:
    # which P row explains the output row best?
    cand = P @ v.double()
    d = (cand - o[r]).norm(dim=1) / o[r].norm()
    best = d.argmin().item()
    # per-d error pattern of row r
    e = (o[r] - ref[r]).abs()
    badd = (e > 0.05 * ref[r].abs().max()).nonzero().flatten().tolist()
    # least squares: o[r] = w @ V  -> effective weights over keys
    w = torch.linalg.lstsq(v.double().T, o[r].unsqueeze(1)).solution.flatten()
    dw = (w - P[r])
    top = dw.abs().topk(8)
    print(f"row {r}: best P row {best} (dist {d[best]:.3f}); bad d count {len(badd)} {badd[:16]}; weight diffs at keys {top.indices.tolist()} = {[round(x,3) for x in top.values.tolist()]}")
