"""Where a K1 launch spends its time OUTSIDE the tile loop: per-workgroup real-time stamps (entry, tile loop start, tile
loop end, output stores done) and the hardware slot each workgroup ran on, from a -DFP_V3_DIAG=3 build of attn_fwd.hip.
  build (CPU box):  DIAG=3 tools/attn_skip_probe.sh build 0      (writes framepack_amd/csrc/probe/libfp_skip0.so)
  run (GPU box):    FRAMEPACK_HIP_LIB=$PWD/framepack_amd/csrc/probe/libfp_skip0.so python tools/attn_timeline.py [S] [heads]
"""
import ctypes
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 19385
H = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
d = H * 128
buf = torch.randn(S, 7 * d, generator=g, device=dev).to(torch.bfloat16)
k, v, q = buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:3 * d]
out = torch.empty(S, d, dtype=torch.bfloat16, device=dev)
for _ in range(20):                                   # warm: the clock settles
    ops.attention(q, k, v, H, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.attention(q, k, v, H, out=out)
b.record()
torch.cuda.synchronize()
L = lib.load()
L.fp_attn_tl_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
nqb = (S + 255) // 256
nwg = nqb * H
ncu = torch.cuda.get_device_properties(0).multi_processor_count
r = nwg % ncu                                         # (attn_launch's policy)
tail = (r if ncu // 8 < r <= ncu // 2 else ncu // 8) if nwg > ncu else 0
split = min(ncu // tail, 8) if tail else 1
grid = nwg - tail + 3 * tail * split if split > 1 else nwg
n = min(grid, 4096)
arr = (ctypes.c_ulonglong * (5 * n))()
assert L.fp_attn_tl_read(arr, n) == 0
rows = [tuple(arr[5 * i:5 * i + 5]) for i in range(n)]
rows = [r for r in rows if r[3] > r[0] > 0]
t0 = min(r[0] for r in rows)
t1 = max(r[3] for r in rows)
TICK = 0.01                                           # us per tick (100 MHz)
print(f"launch {a.elapsed_time(b) * 1e3:.0f} us by events; first entry -> last store {(t1 - t0) * TICK:.0f} us; {len(rows)} workgroups stamped of {grid}")
full = [r for i, r in enumerate(rows) if i < nwg - tail or split == 1]
pro = sorted((r[1] - r[0]) * TICK for r in full)
loop = sorted((r[2] - r[1]) * TICK for r in full)
epi = sorted((r[3] - r[2]) * TICK for r in full)


def stats(x):
    return f"median {x[len(x) // 2]:.1f}  p10 {x[len(x) // 10]:.1f}  p90 {x[len(x) * 9 // 10]:.1f}  max {x[-1]:.1f}"


print("full-sweep workgroups, us:")
print("  entry -> tile loop (Q load, first K/V tiles):", stats(pro))
print("  tile loop:                                   ", stats(loop))
print("  tile loop end -> stores done:                ", stats(epi))
# per hardware slot: (XCC, SE, CU) from HW_ID bits (gfx9: cu_id [11:8], sh_id [12], se_id [15:13]) and XCC_ID [3:0]
slots = defaultdict(list)
for r in rows:
    hw, xcc = r[4] & 0xffffffff, (r[4] >> 32) & 0xf
    slots[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append(r)
gaps, busy, firsts, lasts = [], [], [], []
for key, rs in slots.items():
    rs.sort()
    firsts.append((rs[0][0] - t0) * TICK)
    lasts.append((t1 - rs[-1][3]) * TICK)
    busy.append(sum(r[3] - r[0] for r in rs) * TICK)
    gaps += [(b_[0] - a_[3]) * TICK for a_, b_ in zip(rs, rs[1:])]
gaps.sort()
print(f"{len(slots)} hardware slots (XCC, SE, SH, CU); workgroups per slot: min {min(len(v_) for v_ in slots.values())} max {max(len(v_) for v_ in slots.values())}")
print("  first entry after the launch's first, us:", stats(sorted(firsts)))
print("  idle at the end (last store of the launch - slot's last store), us:", stats(sorted(lasts)))
if gaps:
    print("  gap between consecutive workgroups on a slot, us:", stats(gaps))
span = (t1 - t0) * TICK
print(f"  slot busy (entry -> stores done, summed) / span: mean {sum(busy) / len(busy) / span:.3f}")
print(f"  tile-loop time / span: mean {sum((r[2] - r[1]) for r in rows) * TICK / len(slots) / span:.3f}")
# per XCC: are whole dies slower than others (clock, L2), or single CUs?
by_xcc = defaultdict(list)
for key, rs in slots.items():
    by_xcc[key[0]].append(rs)
for x in sorted(by_xcc):
    loops = sorted((r[2] - r[1]) * TICK for rs in by_xcc[x] for r in rs if (r[2] - r[1]) * TICK > 200)
    ends = sorted((rs[-1][3] - t0) * TICK for rs in by_xcc[x])
    full_end = sorted((max(r[3] for r in rs if (r[2] - r[1]) * TICK > 200) - t0) * TICK for rs in by_xcc[x])
    print(f"  XCC {x}: {len(by_xcc[x])} CUs; full-sweep loop us median {loops[len(loops) // 2]:.1f} (min {loops[0]:.1f} max {loops[-1]:.1f}); "
          f"last FULL sweep ends at {full_end[0]:.0f} .. {full_end[-1]:.0f}; slot's last store at {ends[0]:.0f} .. {ends[-1]:.0f}")
parts = sorted((r[3] - r[0]) * TICK for r in rows if (r[2] - r[1]) * TICK <= 200)
if parts:
    print(f"  {len(parts)} split parts: entry -> stores done us", stats(parts))
