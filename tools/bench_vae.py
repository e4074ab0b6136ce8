"""VAE decode timing at the product shape: 9 latent frames at 80x80 (640x640x33 frames), random weights.
usage: python tools/bench_vae.py [T] [H] [W] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 9
H = int(sys.argv[2]) if len(sys.argv) > 2 else 80
W = int(sys.argv[3]) if len(sys.argv) > 3 else 80
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.manual_seed(0)
vae = AutoencoderKLHunyuanVideo()
for p in vae.parameters():
    if p.dim() > 1:
        torch.nn.init.normal_(p, std=(p[0].numel()) ** -0.5)
vae = vae.to("cuda:0", torch.float16).eval()
z = torch.randn(1, 16, T, H, W, device="cuda:0")
out = vae_decode(z, vae)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters):
    out = vae_decode(z, vae)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
print(f"latents [1,16,{T},{H},{W}] -> {tuple(out.shape)}: {dt * 1e3:.1f} ms per decode, finite={bool(torch.isfinite(out).all())}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
