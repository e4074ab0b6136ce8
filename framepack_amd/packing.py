"""Host-side packing of the FramePack context: RoPE tables, replicate padding, patch extraction.

Integer/index work and the fp32 RoPE tables must be bit-exact (BASELINE.json north_star), so the
tables are built on the HOST with the same torch op sequence the reference uses
(ref diffusers_helper/models/hunyuan_video_packed.py:397-432, 45-60, 822-879) and uploaded once per
section (they are step-invariant; cached by index values).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def pad_to_multiple(x: torch.Tensor, kernel: Tuple[int, int, int]) -> torch.Tensor:
    """Replicate-pad the END of T, H, W up to a multiple of `kernel` (ref :45-51)."""
    t, h, w = x.shape[-3:]
    pads = [(-w) % kernel[2], (-h) % kernel[1], (-t) % kernel[0]]
    return F.pad(x, (0, pads[0], 0, pads[1], 0, pads[2]), mode="replicate")


def _axis_cos_sin(dim: int, pos: torch.Tensor, theta: float):
    # ref :404-408 - pairs (2i, 2i+1) share one frequency (repeat_interleave)
    T, H, W = pos.shape
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    freqs = torch.outer(freqs, pos.reshape(-1)).unflatten(-1, (T, H, W)).repeat_interleave(2, dim=0)
    return freqs.cos(), freqs.sin()


def rope_grid(frame_indices: torch.Tensor, height: int, width: int, axes_dim=(16, 56, 56),
              theta: float = 256.0) -> torch.Tensor:
    """fp32 [B, 256, T, h, w]; channels [cosT, cosY, cosX, sinT, sinY, sinX] (ref :411-432).
    Frame positions are the float values of the index tensor; y, x are arange at the 1x grid."""
    outs = []
    for f in frame_indices.cpu().unbind(0):
        gt, gy, gx = torch.meshgrid(f.to(torch.float32), torch.arange(0, height, dtype=torch.float32),
                                    torch.arange(0, width, dtype=torch.float32), indexing="ij")
        ct, st = _axis_cos_sin(axes_dim[0], gt, theta)
        cy, sy = _axis_cos_sin(axes_dim[1], gy, theta)
        cx, sx = _axis_cos_sin(axes_dim[2], gx, theta)
        outs.append(torch.cat([ct, cy, cx, st, sy, sx], dim=0))
    return torch.stack(outs, dim=0)


def rope_tokens(frame_indices: torch.Tensor, height: int, width: int, pool: int = 1, axes_dim=(16, 56, 56),
                theta: float = 256.0) -> torch.Tensor:
    """[B, tokens, 256] fp32 table of one stream.  pool in {1, 2, 4}: the 2x / 4x context streams
    average-pool the cos/sin VALUES of the full-resolution table after replicate padding
    (ref :857-860, 871-874) - not the positions."""
    g = rope_grid(frame_indices, height, width, axes_dim, theta)
    if pool > 1:
        k = (pool, pool, pool)
        g = F.avg_pool3d(pad_to_multiple(g, k), k, stride=k)
    return g.flatten(2).transpose(1, 2)


class RopeCache:
    """Tables are a pure function of (indices, h, w): build on the host once, keep on the device."""

    def __init__(self, axes_dim, theta, max_entries: int = 8):
        self.axes_dim, self.theta, self.max_entries = tuple(axes_dim), float(theta), max_entries
        self._store: Dict[tuple, torch.Tensor] = {}

    def get(self, streams, height: int, width: int, device) -> torch.Tensor:
        """streams: list of (index tensor [B,T], pool) in final token order -> [B, Si, 256] fp32 on device."""
        key = (tuple((tuple(i.reshape(-1).tolist()), tuple(i.shape), p) for i, p in streams), height, width, str(device))
        hit = self._store.get(key)
        if hit is None:
            parts = [rope_tokens(i, height, width, p, self.axes_dim, self.theta) for i, p in streams]
            hit = torch.cat(parts, dim=1).contiguous().to(device)
            if len(self._store) >= self.max_entries:
                self._store.pop(next(iter(self._store)))
            self._store[key] = hit
        return hit


def patch_rows(x: torch.Tensor, kernel: Tuple[int, int, int]) -> torch.Tensor:
    """[B,C,T,H,W] -> [B*tokens, C*pt*ph*pw]: the im2col of a Conv3d whose stride equals its kernel
    (ref :669, 675-677), token order (t, h, w), feature order (c, pt, ph, pw) = Conv3d weight order."""
    B, C, T, H, W = x.shape
    pt, ph, pw = kernel
    x = x.view(B, C, T // pt, pt, H // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(B * (T // pt) * (H // ph) * (W // pw), C * pt * ph * pw)


def unpatchify(tokens: torch.Tensor, B: int, t: int, h: int, w: int, c: int, pt: int, p: int) -> torch.Tensor:
    """'b (t h w) (c pt ph pw) -> b c (t pt) (h ph) (w pw)'  (ref :1055-1057)."""
    x = tokens.view(B, t, h, w, c, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7)
    return x.reshape(B, c, t * pt, h * p, w * p)


def get_cu_seqlens(text_mask: torch.Tensor, img_len: int, device=None) -> torch.Tensor:
    """int32 [2B+1]: per sample the valid tokens (img + text_len) and the padding tail are two varlen
    segments (ref :63-77, which hard-codes device='cuda')."""
    B, L = text_mask.shape
    text_len = text_mask.sum(dim=1).tolist()
    max_len = L + img_len
    cu = [0]
    for i in range(B):
        cu.append(i * max_len + int(text_len[i]) + img_len)
        cu.append((i + 1) * max_len)
    return torch.tensor(cu, dtype=torch.int32, device=device)
