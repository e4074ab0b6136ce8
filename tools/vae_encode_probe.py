"""Full-size VAE encode of one 640x640 image (random weights): time and memory. usage: vae_encode_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode, vae_encode  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
vae = AutoencoderKLHunyuanVideo().to(dev, torch.float16).eval()
for p in vae.parameters():
    p.data.mul_(0.5)
x = torch.rand(1, 3, 1, 640, 640, device=dev) * 2 - 1
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z = vae_encode(x, vae)
    torch.cuda.synchronize()
    print(f"encode {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms, latent {tuple(z.shape)}, finite {bool(torch.isfinite(z).all())}, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
