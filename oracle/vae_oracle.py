"""Functional CPU restatement of the causal-3D VAE: decode, and the encode half behind `vae_encode`.

TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED.  The reference's own code for this path is six lines
(ref: diffusers_helper/hunyuan.py:93-103 `vae_decode`: latents / scaling_factor ->
vae.dtype -> `vae.decode(latents).sample`); everything below that call lives in
diffusers==0.33.1 `AutoencoderKLHunyuanVideo` (requirements.txt:2; constructed at
ref: framepack/models.py:158-163), which is absent from this image, and the reference
holds no test, fixture or stored output for it.  The functions here restate that class
from its published algorithm (each marked `[diffusers, unpinned]`); nothing in this
container can confirm them.  What the GPU tests therefore establish is that the HIP
decoder computes THIS algorithm; the first time the real package or real VAE weights
are available, `strict` state-dict loading and a decode comparison must re-check it.

The encode half (ref: diffusers_helper/hunyuan.py:107-110 `vae_encode`: `vae.encode(image).latent_dist.sample()`
times scaling_factor; called on single images, ref: framepack/worker.py:246-272) is restated under the same terms.

`P` is a flat state dict with diffusers' parameter names (`decoder.conv_in.conv.weight`,
`decoder.up_blocks.0.resnets.0.norm1.weight`, `post_quant_conv.weight`, ...), volumes
are NCDHW as in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class VAEConfig:
    """[diffusers, unpinned] AutoencoderKLHunyuanVideo ctor defaults = hunyuanvideo-community/HunyuanVideo vae."""
    latent_channels: int = 16
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.476986
    spatial_compression_ratio: int = 8
    temporal_compression_ratio: int = 4
    mid_block_add_attention: bool = True
    # framewise decoding knobs (use_framewise_decoding is on by default in diffusers)
    tile_sample_min_num_frames: int = 16
    tile_sample_stride_num_frames: int = 12


def up_block_plan(cfg: VAEConfig) -> List[Tuple[int, int, Tuple[int, int, int] | None]]:
    """[diffusers, unpinned] HunyuanVideoDecoder3D.__init__: (in_ch, out_ch, upsample factor or None) per up block.
    With ratios 8 / 4 and four blocks: block 0 upsamples (1,2,2), blocks 1 and 2 (2,2,2), block 3 nothing."""
    import math
    rev = list(reversed(cfg.block_out_channels))
    n = len(rev)
    n_sp = int(math.log2(cfg.spatial_compression_ratio))
    n_t = int(math.log2(cfg.temporal_compression_ratio))
    plan = []
    out_ch = rev[0]
    for i in range(n):
        prev, out_ch = out_ch, rev[i]
        final = i == n - 1
        sp = i < n_sp
        tm = (i >= n - 1 - n_t) and not final
        factor = ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1))
        plan.append((prev, out_ch, factor if (sp or tm) else None))
    return plan


def causal_conv3d(P: Dict[str, Tensor], name: str, x: Tensor, stride: Tuple[int, int, int] = (1, 1, 1)) -> Tensor:
    """[diffusers, unpinned] HunyuanVideoCausalConv3d: replicate padding, (k-1) frames in front and none
    behind in time, k//2 on each side in space; then a plain (optionally strided) Conv3d (weights at `<name>.conv`)."""
    w = P[name + ".conv.weight"]
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0), mode="replicate")
    return F.conv3d(x, w, P.get(name + ".conv.bias"), stride=stride)


def group_norm(P: Dict[str, Tensor], name: str, x: Tensor, groups: int, eps: float = 1e-6) -> Tensor:
    return F.group_norm(x, groups, P[name + ".weight"], P[name + ".bias"], eps)


def resnet_block(P: Dict[str, Tensor], name: str, x: Tensor, groups: int) -> Tensor:
    """[diffusers, unpinned] HunyuanVideoResnetBlockCausal3D (eps 1e-6, swish, dropout 0)."""
    h = F.silu(group_norm(P, name + ".norm1", x, groups))
    h = causal_conv3d(P, name + ".conv1", h)
    h = F.silu(group_norm(P, name + ".norm2", h, groups))
    h = causal_conv3d(P, name + ".conv2", h)
    if (name + ".conv_shortcut.conv.weight") in P:
        x = causal_conv3d(P, name + ".conv_shortcut", x)
    return h + x


def frame_causal_mask(num_frames: int, hw: int, dtype: torch.dtype) -> Tensor:
    """[diffusers, unpinned] prepare_causal_attention_mask: token i may attend to token j iff frame(j) <= frame(i)."""
    f = torch.arange(num_frames).repeat_interleave(hw)
    allowed = f[None, :] <= f[:, None]
    return torch.zeros(num_frames * hw, num_frames * hw, dtype=dtype).masked_fill(~allowed, float("-inf"))


def mid_attention(P: Dict[str, Tensor], name: str, x: Tensor, groups: int) -> Tensor:
    """[diffusers, unpinned] the mid block's Attention (one head of C channels, GroupNorm inside, residual
    connection, AttnProcessor2_0) applied to the volume flattened to tokens in (t, h, w) order."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).flatten(1, 3)                          # [B, S, C]
    res = tok
    hn = F.group_norm(tok.transpose(1, 2), groups, P[name + ".group_norm.weight"], P[name + ".group_norm.bias"],
                      1e-6).transpose(1, 2)
    q = F.linear(hn, P[name + ".to_q.weight"], P[name + ".to_q.bias"])
    k = F.linear(hn, P[name + ".to_k.weight"], P[name + ".to_k.bias"])
    v = F.linear(hn, P[name + ".to_v.weight"], P[name + ".to_v.bias"])
    mask = frame_causal_mask(T, H * W, q.dtype)
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], attn_mask=mask)[:, 0]
    o = F.linear(o, P[name + ".to_out.0.weight"], P[name + ".to_out.0.bias"]) + res
    return o.unflatten(1, (T, H, W)).permute(0, 4, 1, 2, 3)


def upsample_causal(P: Dict[str, Tensor], name: str, x: Tensor, factor: Tuple[int, int, int]) -> Tensor:
    """[diffusers, unpinned] HunyuanVideoUpsampleCausal3D: the first frame is upsampled in space only, the
    others by the full factor (nearest), then a causal conv."""
    first, rest = x[:, :, :1], x[:, :, 1:]
    first = F.interpolate(first.squeeze(2), scale_factor=factor[1:], mode="nearest").unsqueeze(2)
    if rest.shape[2] > 0:
        rest = F.interpolate(rest.contiguous(), scale_factor=factor, mode="nearest")
        x = torch.cat([first, rest], dim=2)
    else:
        x = first
    return causal_conv3d(P, name + ".conv", x)


def decoder(P: Dict[str, Tensor], cfg: VAEConfig, z: Tensor) -> Tensor:
    """[diffusers, unpinned] HunyuanVideoDecoder3D.forward."""
    g = cfg.norm_num_groups
    h = causal_conv3d(P, "decoder.conv_in", z)
    h = resnet_block(P, "decoder.mid_block.resnets.0", h, g)
    if cfg.mid_block_add_attention:
        h = mid_attention(P, "decoder.mid_block.attentions.0", h, g)
    h = resnet_block(P, "decoder.mid_block.resnets.1", h, g)
    for i, (_, _, factor) in enumerate(up_block_plan(cfg)):
        for j in range(cfg.layers_per_block + 1):
            h = resnet_block(P, f"decoder.up_blocks.{i}.resnets.{j}", h, g)
        if factor is not None:
            h = upsample_causal(P, f"decoder.up_blocks.{i}.upsamplers.0", h, factor)
    h = F.silu(group_norm(P, "decoder.conv_norm_out", h, g))
    return causal_conv3d(P, "decoder.conv_out", h)


def decode_tile(P: Dict[str, Tensor], cfg: VAEConfig, z: Tensor) -> Tensor:
    """post_quant_conv (1x1x1) then the decoder."""
    z = F.conv3d(z, P["post_quant_conv.weight"], P["post_quant_conv.bias"])
    return decoder(P, cfg, z)


def blend_t(a: Tensor, b: Tensor, extent: int) -> Tensor:
    """[diffusers, unpinned] AutoencoderKLHunyuanVideo.blend_t (linear cross-fade of the first `extent` frames)."""
    extent = min(a.shape[2], b.shape[2], extent)
    b = b.clone()
    for x in range(extent):
        b[:, :, x] = a[:, :, -extent + x] * (1 - x / extent) + b[:, :, x] * (x / extent)
    return b


def decode(P: Dict[str, Tensor], cfg: VAEConfig, z: Tensor, framewise: bool = True) -> Tensor:
    """[diffusers, unpinned] AutoencoderKLHunyuanVideo._decode with spatial tiling off (the reference only
    enables tiling/slicing on low-VRAM GPUs, ref: framepack/models.py:280-282) and framewise decoding on
    (diffusers' default): more than 4 latent frames are decoded in windows of 4+1 latent frames at stride 3
    and cross-faded over 4 output frames (_temporal_tiled_decode)."""
    T = z.shape[2]
    t_min = cfg.tile_sample_min_num_frames // cfg.temporal_compression_ratio
    t_stride = cfg.tile_sample_stride_num_frames // cfg.temporal_compression_ratio
    if not framewise or T <= t_min:
        return decode_tile(P, cfg, z)
    n_out = (T - 1) * cfg.temporal_compression_ratio + 1
    blend = cfg.tile_sample_min_num_frames - cfg.tile_sample_stride_num_frames
    rows = []
    for i in range(0, T, t_stride):
        d = decode_tile(P, cfg, z[:, :, i:i + t_min + 1])
        if i > 0:
            d = d[:, :, 1:]
        rows.append(d)
    out = []
    for i, tile in enumerate(rows):
        if i > 0:
            tile = blend_t(rows[i - 1], tile, blend)
            out.append(tile[:, :, :cfg.tile_sample_stride_num_frames])
        else:
            out.append(tile[:, :, :cfg.tile_sample_stride_num_frames + 1])
    return torch.cat(out, dim=2)[:, :, :n_out]


def vae_decode(P: Dict[str, Tensor], cfg: VAEConfig, latents: Tensor, framewise: bool = True) -> Tensor:
    """ref: diffusers_helper/hunyuan.py:93-103 (image_mode=False branch)."""
    return decode(P, cfg, latents / cfg.scaling_factor, framewise)


# ---------------------------------------------------------------------------------------------------- encode half
def down_block_plan(cfg: VAEConfig) -> List[Tuple[int, int, Tuple[int, int, int] | None]]:
    """[diffusers, unpinned] HunyuanVideoEncoder3D.__init__: (in_ch, out_ch, downsample stride or None) per down
    block.  With ratios 8 / 4 and four blocks: block 0 strides (1,2,2), blocks 1 and 2 (2,2,2), block 3 nothing."""
    import math
    ch = cfg.block_out_channels
    n = len(ch)
    n_sp = int(math.log2(cfg.spatial_compression_ratio))
    n_t = int(math.log2(cfg.temporal_compression_ratio))
    plan = []
    out_ch = ch[0]
    for i in range(n):
        prev, out_ch = out_ch, ch[i]
        final = i == n - 1
        sp = i < n_sp
        tm = (i >= n - 1 - n_t) and not final
        plan.append((prev, out_ch, ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None))
    return plan


def encoder(P: Dict[str, Tensor], cfg: VAEConfig, x: Tensor) -> Tensor:
    """[diffusers, unpinned] HunyuanVideoEncoder3D.forward: conv_in, down blocks (resnets, then a strided causal
    conv), mid block, GroupNorm + SiLU, conv_out to 2 x latent channels."""
    g = cfg.norm_num_groups
    h = causal_conv3d(P, "encoder.conv_in", x)
    for i, (_, _, stride) in enumerate(down_block_plan(cfg)):
        for j in range(cfg.layers_per_block):
            h = resnet_block(P, f"encoder.down_blocks.{i}.resnets.{j}", h, g)
        if stride is not None:
            h = causal_conv3d(P, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride)
    h = resnet_block(P, "encoder.mid_block.resnets.0", h, g)
    if cfg.mid_block_add_attention:
        h = mid_attention(P, "encoder.mid_block.attentions.0", h, g)
    h = resnet_block(P, "encoder.mid_block.resnets.1", h, g)
    h = F.silu(group_norm(P, "encoder.conv_norm_out", h, g))
    return causal_conv3d(P, "encoder.conv_out", h)


def encode_moments(P: Dict[str, Tensor], cfg: VAEConfig, x: Tensor) -> Tuple[Tensor, Tensor]:
    """[diffusers, unpinned] AutoencoderKLHunyuanVideo.encode without tiling (inputs of at most 16 frames; the
    reference encodes single images): encoder, quant_conv (1x1x1), then DiagonalGaussianDistribution's
    (mean, logvar clamped to [-30, 20])."""
    m = F.conv3d(encoder(P, cfg, x), P["quant_conv.weight"], P["quant_conv.bias"])
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_encode(P: Dict[str, Tensor], cfg: VAEConfig, image: Tensor, noise: Tensor | None) -> Tensor:
    """ref: diffusers_helper/hunyuan.py:107-110.  `latent_dist.sample()` = mean + exp(logvar / 2) * noise; the noise
    is an argument here (None = the mode) so that a test can feed both sides the same draw."""
    mean, logvar = encode_moments(P, cfg, image)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    return z * cfg.scaling_factor


def random_state_dict(cfg: VAEConfig, seed: int = 0, dtype: torch.dtype = torch.float32,
                      with_encoder: bool = False, in_channels: int = 3) -> Dict[str, Tensor]:
    """Seeded random weights with diffusers' names (no checkpoint exists offline).  The encoder half is drawn
    after the decoder half, so the decoder weights of a seed do not depend on `with_encoder`."""
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, Tensor] = {}

    def conv(name: str, cin: int, cout: int, k: int) -> None:
        fan = cin * k * k * k
        P[name + ".weight"] = (torch.randn(cout, cin, k, k, k, generator=g) / fan ** 0.5).to(dtype)
        P[name + ".bias"] = (torch.randn(cout, generator=g) * 0.02).to(dtype)

    def norm(name: str, c: int) -> None:
        P[name + ".weight"] = (1 + 0.1 * torch.randn(c, generator=g)).to(dtype)
        P[name + ".bias"] = (0.1 * torch.randn(c, generator=g)).to(dtype)

    def lin(name: str, cin: int, cout: int) -> None:
        P[name + ".weight"] = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dtype)
        P[name + ".bias"] = (torch.randn(cout, generator=g) * 0.02).to(dtype)

    def resnet(name: str, cin: int, cout: int) -> None:
        norm(name + ".norm1", cin)
        conv(name + ".conv1.conv", cin, cout, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2.conv", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut.conv", cin, cout, 1)

    top = cfg.block_out_channels[-1]
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    conv("decoder.conv_in.conv", cfg.latent_channels, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(a + "." + n, top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    for i, (cin, cout, factor) in enumerate(up_block_plan(cfg)):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if factor is not None:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv.conv", cout, cout, 3)
    norm("decoder.conv_norm_out", cfg.block_out_channels[0])
    conv("decoder.conv_out.conv", cfg.block_out_channels[0], cfg.out_channels, 3)
    if with_encoder:
        conv("encoder.conv_in.conv", in_channels, cfg.block_out_channels[0], 3)
        for i, (cin, cout, stride) in enumerate(down_block_plan(cfg)):
            for j in range(cfg.layers_per_block):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
            if stride is not None:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv.conv", cout, cout, 3)
        resnet("encoder.mid_block.resnets.0", top, top)
        a = "encoder.mid_block.attentions.0"
        norm(a + ".group_norm", top)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(a + "." + n, top, top)
        resnet("encoder.mid_block.resnets.1", top, top)
        norm("encoder.conv_norm_out", top)
        conv("encoder.conv_out.conv", top, 2 * cfg.latent_channels, 3)
        conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return P
