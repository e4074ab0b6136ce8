"""Randomised check of K11 (causal 3x3x3 convolution: folded upsampling, strides, residual epilogue, narrow tiles, Cin
that is / is not a power of two) against the CPU oracle's convolution.  Not part of the test suite; run on the GPU box:
python tools/conv_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import ops  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402  (the checker; tools are not product code)

dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
worst = 0.0
for case in range(cases):
    g = torch.Generator().manual_seed(case)
    T, H, W = rnd.randint(1, 5), rnd.randint(1, 24), rnd.randint(1, 24)
    cin = rnd.choice([64, 64, 128, 128, 192, 256])
    big = rnd.random() < 0.2          # frames of 32 - 64 whole row tiles: the band-major tile walk (pick_band, vae_kernels.hip)
    if big:
        T = rnd.randint(2, 4)
        H, W = rnd.choice([(64, 128), (48, 192), (80, 128), (68, 128), (96, 96), (128, 128), (32, 256)])
        cin = rnd.choice([128, 256])
    cout = 8 * rnd.choice([1, 4, 8, 9, 16, 17, 32])
    mode = rnd.choice(["plain", "plain", "up", "stride"])
    if big and mode == "up":          # (the upsampled frame is the one that has to hold whole tiles)
        H, W = H // 2, W // 2
    x = torch.randn(T, H, W, cin, generator=g).half()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * (27 * cin) ** -0.5).half()
    b = (torch.randn(cout, generator=g) * 0.1).half()
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, -1).contiguous().to(dev)
    xr = x.float().permute(3, 0, 1, 2).unsqueeze(0)
    if mode == "stride":
        st = rnd.choice([(1, 2, 2), (2, 2, 2), (2, 1, 1)])
        ref = VO.causal_conv3d({"c.conv.weight": w.float(), "c.conv.bias": b.float()}, "c", xr, st)[0].permute(1, 2, 3, 0)
        got = ops.conv3d_causal_strided(x.to(dev), wk, b.to(dev), st)
        desc = f"stride {st}"
    else:
        up = rnd.choice([(1, 2, 2), (2, 2, 2)]) if mode == "up" else (1, 1, 1)
        if up != (1, 1, 1):
            ref = VO.upsample_causal({"u.conv.conv.weight": w.float(), "u.conv.conv.bias": b.float()}, "u", xr, up)
        else:
            ref = VO.causal_conv3d({"c.conv.weight": w.float(), "c.conv.bias": b.float()}, "c", xr)
        ref = ref[0].permute(1, 2, 3, 0)
        res = torch.randn(*ref.shape, generator=g).half() if rnd.random() < 0.5 else None
        if res is not None:
            ref = ref + res.float()
        got = ops.conv3d_causal(x.to(dev), wk, b.to(dev), up=up, resid=None if res is None else res.to(dev))
        desc = f"up {up} resid {res is not None}"
    err = ((got.double().cpu() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30)).item()
    worst = max(worst, err)
    if tuple(got.shape) != tuple(ref.shape) or not err < 3e-3:
        print(f"FAIL case {case}: T,H,W={T},{H},{W} cin={cin} cout={cout} {desc}: rel-L2 {err:.3e}")
        sys.exit(1)
print(f"{cases} cases passed; worst rel-L2 {worst:.2e}", flush=True)

# The 512 x 128 tile (Cout = 128 over two rounds of tiles or more, gemm_w4.h TALL) against the 256 x 128 tile that the
# cases above pin to the oracle: the same bits (same K order per output element).  Random volumes of 270 000 - 600 000
# voxels on the device: ragged last tiles, frames that are / are not whole tiles (the band walk), Cin 128 / 256, upsampling.
from framepack_amd import lib as fplib  # noqa: E402
ncu = torch.cuda.get_device_properties(0).multi_processor_count
tall_cases = max(4, cases // 6)
for case in range(tall_cases):
    gd = torch.Generator(device=dev).manual_seed(1000 + case)
    cin = rnd.choice([128, 128, 256])
    up = rnd.choice([(1, 1, 1), (1, 1, 1), (1, 2, 2), (2, 2, 2)])
    while True:
        T, H, W = rnd.randint(1, 4), rnd.randint(100, 420), rnd.choice([rnd.randint(100, 420), 256, 384, 512])
        To = 2 * T - 1 if up[0] == 2 else T
        M = To * H * up[1] * W * up[2]
        if 2 * 512 * ncu <= M <= 600000:
            break
    x = torch.randn(T, H, W, cin, generator=gd, device=dev).half()
    wk = (torch.randn(128, 27 * cin, generator=gd, device=dev) * (27 * cin) ** -0.5).half()
    b = (torch.randn(128, generator=gd, device=dev) * 0.1).half()
    res = torch.randn(To, H * up[1], W * up[2], 128, generator=gd, device=dev).half() if rnd.random() < 0.5 else None
    tall = ops.conv3d_causal(x, wk, b, up=up, resid=res)
    old = fplib.load().fp_debug_gemm_variant(1 | 4)
    narrow = ops.conv3d_causal(x, wk, b, up=up, resid=res)
    fplib.load().fp_debug_gemm_variant(old)
    if not (torch.isfinite(tall.float()).all() and torch.equal(tall, narrow)):
        print(f"FAIL tall case {case}: T,H,W={T},{H},{W} cin={cin} up {up} resid {res is not None}: "
              f"max diff {(tall.float() - narrow.float()).abs().max().item():.3e}")
        sys.exit(1)
print(f"{tall_cases} tall-tile cases bit-identical to the narrow tile")
