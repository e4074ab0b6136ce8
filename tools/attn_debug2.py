import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402
L = lib.load()
dev = "cuda:0"
S, H = 64, 1
g = torch.Generator("cpu").manual_seed(S)
q, k, v = (torch.randn(S, H * 128, generator=g).to(dev).bfloat16() for _ in range(3))
sc = (q.double() @ k.double().T) / math.sqrt(128.0) * 1.4426950408889634
L.fp_debug_attn_variant(1)
o = ops.attention(q, k, v, H)
torch.cuda.synchronize()
dbg = o.view(torch.uint8).view(S, -1)[:, :32].contiguous().view(torch.float32)
mref = sc.max(dim=1).values
lref = torch.exp2(sc - mref[:, None]).sum(dim=1)
for r in range(12):
    print(r, "l_tot %.4f m %.4f l_run %.4f p0 %.4f p1 %.4f o0 %.4f s0 %.4f s1 %.4f" % tuple(dbg[r].tolist()),
          "| ref m %.4f l %.4f s[r][0] %.4f s[r][1] %.4f" % (mref[r].item(), lref[r].item(), sc[r, 0].item() / 1.4426950408889634 * math.sqrt(128), sc[r, 1].item() / 1.4426950408889634 * math.sqrt(128)))
