"""Per-segment cycle shares of the attention tile loop (stamped diagnostic build).
Build first:  make -C framepack_amd/csrc clean && make -C framepack_amd/csrc DIAG=1 ; rebuild without DIAG afterwards."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops, lib
S, H = 19385, 24
d = H * 128
g = torch.Generator(device="cuda:0").manual_seed(0)
q, k, v = (torch.randn(S, d, generator=g, device="cuda:0").to(torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
for _ in range(3):
    ops.attention(q, k, v, H, out=out)
torch.cuda.synchronize()
L = lib.load()
buf = (ctypes.c_ulonglong * 64)()
print("rc", L.fp_attn_dbg_read(buf))
for w in range(8):
    a = [buf[w * 8 + i] for i in range(8)]
    nt = max(a[5], 1)
    print(f"wave {w}: per tile cycles  loads {a[6]/nt:6.0f} max {(a[7]>>32)/nt:6.0f} rescale {(a[7]&0xffffffff)/nt:6.0f}  exp {a[0]/nt:7.0f}  waitB1 {a[1]/nt:7.0f}  M {a[2]/nt:7.0f}  W {a[3]/nt:7.0f}  waitB2 {a[4]/nt:7.0f}  total {sum(a[:5])/nt:7.0f}")
