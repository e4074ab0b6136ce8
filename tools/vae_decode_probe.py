"""One section's VAE decode (latent [1,16,9,80,80] -> 33 frames 640x640, random weights) a few times: for rocprofv3
kernel stats.  usage: vae_decode_probe.py [iters]
GEMM_AB=1,5: alternate fp_debug_gemm_variant decode by decode (5 = 1 | 4: the 256 x 128 tile for the Cout = 128
convolutions instead of the 512 x 128 one), same process, same box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
vae = AutoencoderKLHunyuanVideo().to(dev, torch.float16).eval()
for p in vae.parameters():
    p.data.mul_(0.5)
z = torch.randn(1, 16, 9, 80, 80, device=dev)
AB = [int(v) for v in os.environ.get("GEMM_AB", "").split(",") if v]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    if AB:
        from framepack_amd import lib as _fplib
        _fplib.load().fp_debug_gemm_variant(AB[i % len(AB)])
        print(f"variant {AB[i % len(AB)]}:", end=" ")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vae_decode(z, vae)
    torch.cuda.synchronize()
    print(f"decode {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms {tuple(out.shape)}", flush=True)
