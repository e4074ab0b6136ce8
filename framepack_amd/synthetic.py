"""Synthetic configurations, weights and inputs for the packed DiT (BASELINE.json configs).

Pure host-side tensor construction: no kernels, no oracle.  Shapes and seeds follow
SURVEY.md 8(d); the index layout follows ref framepack/worker.py:463-465.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

# BASELINE.json configs[0]: tiny plumbing model (2 double + 2 single blocks, hidden 256 = 2 heads x 128)
CONFIG_TINY = dict(in_channels=16, out_channels=16, num_attention_heads=2, attention_head_dim=128,
                   num_layers=2, num_single_layers=2, num_refiner_layers=2, mlp_ratio=4.0, patch_size=2,
                   patch_size_t=1, qk_norm="rms_norm", guidance_embeds=True, text_embed_dim=4096,
                   pooled_projection_dim=768, rope_theta=256.0, rope_axes_dim=(16, 56, 56),
                   has_image_proj=True, image_proj_dim=1152, has_clean_x_embedder=True)

# BASELINE.json configs[1]: HunyuanVideo-13B FramePack I2V (ref hunyuan_video_packed.py:700-721 defaults
# + image projection + clean-latent embedder as lllyasviel/FramePackI2V_HY ships them)
CONFIG_13B = dict(CONFIG_TINY, num_attention_heads=24, num_layers=20, num_single_layers=40)


def init_synthetic_(named_tensors: Dict[str, torch.Tensor], seed: int = 0, bias_std: float = 0.02,
                    fast_device_rng: bool = False) -> None:
    """Fill parameters in sorted-name order: matrices N(0, 1/fan_in), biases N(0, bias_std^2),
    1-D norm weights 1 + N(0, 0.1^2).  With fast_device_rng the draw happens on the tensor's own
    device (13B bench: values differ from the CPU stream, distribution is the same)."""
    g = None if fast_device_rng else torch.Generator("cpu").manual_seed(seed)
    if fast_device_rng:
        torch.manual_seed(seed)
    for name in sorted(named_tensors):
        t = named_tensors[name]
        shape = tuple(t.shape)

        def draw():
            if fast_device_rng:
                return torch.randn(shape, device=t.device, dtype=torch.float32)
            return torch.randn(shape, generator=g)

        if name.endswith(".bias"):
            v = draw() * bias_std
        elif len(shape) == 1:
            v = 1.0 + 0.1 * draw()
        else:
            v = draw() * (1.0 / math.sqrt(int(np.prod(shape[1:]))))
        with torch.no_grad():
            t.copy_(v.to(device=t.device, dtype=t.dtype))


def section_indices(latent_window_size: int = 9, latent_padding: int = 0):
    """ref framepack/worker.py:463-465: arange(1+pad+W+1+2+16) split into
    clean_pre | blank | latent | clean_post | 2x | 4x; clean = cat(pre, post)."""
    pad = latent_padding * latent_window_size
    idx = torch.arange(0, 1 + pad + latent_window_size + 1 + 2 + 16).unsqueeze(0)
    pre, _blank, lat, post, i2, i4 = idx.split([1, pad, latent_window_size, 1, 2, 16], dim=1)
    return dict(latent_indices=lat, clean_latent_indices=torch.cat([pre, post], dim=1),
                clean_latent_2x_indices=i2, clean_latent_4x_indices=i4)


def make_inputs(seed: int = 31337, frames_latent: int = 9, height: int = 640, width: int = 640,
                n_text: int = 256, text_pad: int = 512, text_dim: int = 4096, pooled_dim: int = 768,
                image_tokens: int = 729, image_dim: int = 1152, latent_padding: int = 0,
                sigma: float = 1.0, guidance_scale: float = 10.0, dtype=torch.bfloat16,
                with_2x: bool = True, with_4x: bool = True) -> Dict[str, Optional[torch.Tensor]]:
    """Keyword arguments for `forward` (ref hunyuan_video_packed.py:881-890), on CPU.
    Latents are drawn exactly as ref k_diffusion_hunyuan.py:63; clean latents are fp32 CPU
    tensors and indices int64 CPU, as the worker passes them."""
    g = torch.Generator("cpu").manual_seed(seed)
    h8, w8 = height // 8, width // 8
    latents = torch.randn((1, 16, frames_latent, h8, w8), generator=g)
    kw = section_indices(frames_latent, latent_padding)
    kw["hidden_states"] = latents.to(dtype)
    kw["timestep"] = torch.tensor([sigma * 1000.0]).to(dtype)
    kw["guidance"] = torch.tensor([guidance_scale * 1000.0]).to(dtype)
    kw["encoder_hidden_states"] = torch.randn((1, text_pad, text_dim), generator=g).to(dtype)
    mask = torch.zeros((1, text_pad), dtype=torch.bool)
    mask[:, :n_text] = True
    kw["encoder_attention_mask"] = mask
    kw["pooled_projections"] = torch.randn((1, pooled_dim), generator=g).to(dtype)
    kw["image_embeddings"] = torch.randn((1, image_tokens, image_dim), generator=g).to(dtype)
    kw["clean_latents"] = torch.randn((1, 16, 2, h8, w8), generator=g)
    if with_2x:
        kw["clean_latents_2x"] = torch.randn((1, 16, 2, h8, w8), generator=g)
    else:
        kw["clean_latents_2x"] = None
        kw["clean_latent_2x_indices"] = None
    if with_4x:
        kw["clean_latents_4x"] = torch.randn((1, 16, 16, h8, w8), generator=g)
    else:
        kw["clean_latents_4x"] = None
        kw["clean_latent_4x_indices"] = None
    return kw


def packed_lengths(frames_latent: int, height: int, width: int, n_text: int, image_tokens: int = 729,
                   with_2x: bool = True, with_4x: bool = True):
    """(Si, St, S) for a section: SURVEY.md 8 canonical-shape table."""
    h, w = height // 16, width // 16
    si = frames_latent * h * w + 2 * h * w
    if with_2x:
        si += 1 * math.ceil(h / 2) * math.ceil(w / 2)
    if with_4x:
        si += 4 * math.ceil(h / 4) * math.ceil(w / 4)
    st = image_tokens + n_text
    return si, st, si + st
