"""Functional CPU restatement of HunyuanVideoTransformer3DModelPacked.forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All `ref:` citations are paths
relative to /root/reference.  `P` is a flat state dict with the reference's
parameter names; `cfg` is a DiTConfig.  Every function follows the reference's
cast points, so running it with bf16 parameters reproduces the reference's bf16
rounding, and running it in fp32 gives the "truth" the GPU path is scored
against.

Third-party pieces (diffusers==0.33.1, absent from the image) are restated from
the published algorithm and marked `[diffusers, unpinned]`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class DiTConfig:
    """ref: diffusers_helper/models/hunyuan_video_packed.py:700-721 (ctor defaults)."""
    in_channels: int = 16
    out_channels: int = 16
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    num_layers: int = 20
    num_single_layers: int = 40
    num_refiner_layers: int = 2
    mlp_ratio: float = 4.0
    patch_size: int = 2
    patch_size_t: int = 1
    qk_norm: str = "rms_norm"
    guidance_embeds: bool = True
    text_embed_dim: int = 4096
    pooled_projection_dim: int = 768
    rope_theta: float = 256.0
    rope_axes_dim: Tuple[int, int, int] = (16, 56, 56)
    has_image_proj: bool = False
    image_proj_dim: int = 1152
    has_clean_x_embedder: bool = False

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------

def linear(P: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    """nn.Linear; if `<name>.scale_weight` exists the weight is FP8 and is
    dequantised first.  ref: utils/fp8_optimization_utils.py:216-226."""
    w = P[name + ".weight"]
    b = P.get(name + ".bias")
    sk = name + ".scale_weight"
    if sk in P:
        w = w.to(P[sk].dtype) * P[sk]
    return F.linear(x, w, b)


def layer_norm(x: Tensor, weight=None, bias=None, eps: float = 1e-6) -> Tensor:
    """ref: diffusers_helper/dit_common.py:9-10 (F.layer_norm in the input dtype, `.to(x)`)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps).to(x)


def rms_norm(x: Tensor, weight: Optional[Tensor], eps: float = 1e-6) -> Tensor:
    """ref: diffusers_helper/dit_common.py:31-39."""
    dt = x.dtype
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)
    if weight is None:
        return y.to(dt)
    return y.to(dt) * weight.to(dt)


def sinusoid_256(t: Tensor) -> Tensor:
    """[diffusers, unpinned] Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0):
    fp32 [cos | sin] of t * 10000^(-i/128).  Call sites: ref hunyuan_video_packed.py:195,219."""
    half = 128
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    ang = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def mlp_silu(P, name: str, x: Tensor) -> Tensor:
    """[diffusers, unpinned] TimestepEmbedding / PixArtAlphaTextProjection(act_fn='silu'):
    linear_2(silu(linear_1(x)))."""
    return linear(P, name + ".linear_2", F.silu(linear(P, name + ".linear_1", x)))


def sdpa_nhd(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """ref: hunyuan_video_packed.py:97 - non-causal SDPA, default scale, [B,S,H,D] layout."""
    return F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)


def attention_reference_f64(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """Independent float64 softmax(QK^T/sqrt(D))V for [S,H,D] inputs (kernel tests)."""
    qd, kd, vd = (t.to(torch.float64).transpose(0, 1) for t in (q, k, v))     # [H,S,D]
    s = qd @ kd.transpose(1, 2) / math.sqrt(q.shape[-1])
    p = torch.softmax(s, dim=-1)
    return (p @ vd).transpose(0, 1)


# ----------------------------------------------------------------------------
# packing: RoPE tables, padding, patch embeds
# ----------------------------------------------------------------------------

def pad_for_3d_conv(x: Tensor, kernel: Tuple[int, int, int]) -> Tensor:
    """ref: hunyuan_video_packed.py:45-51 - replicate-pad the END of T/H/W to a kernel multiple."""
    _, _, t, h, w = x.shape
    pt, ph, pw = kernel
    return F.pad(x, (0, (pw - w % pw) % pw, 0, (ph - h % ph) % ph, 0, (pt - t % pt) % pt), mode="replicate")


def center_down_sample_3d(x: Tensor, kernel) -> Tensor:
    """ref: hunyuan_video_packed.py:54-60 - avg_pool3d of the VALUES (cos/sin), not of positions."""
    return F.avg_pool3d(x, kernel, stride=kernel)


def rope_axis(dim: int, pos: Tensor, theta: float) -> Tuple[Tensor, Tensor]:
    """ref: hunyuan_video_packed.py:404-408."""
    T, H, W = pos.shape
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=pos.device)[: dim // 2] / dim))
    freqs = torch.outer(freqs, pos.reshape(-1)).unflatten(-1, (T, H, W)).repeat_interleave(2, dim=0)
    return freqs.cos(), freqs.sin()


def rope_table(cfg: DiTConfig, frame_indices: Tensor, height: int, width: int) -> Tensor:
    """ref: hunyuan_video_packed.py:411-432.  frame_indices [B,T] -> fp32 [B,256,T,h,w],
    channel order [cosT,cosY,cosX,sinT,sinY,sinX] (:423)."""
    DT, DY, DX = cfg.rope_axes_dim
    out = []
    for f in frame_indices.unbind(0):
        GT, GY, GX = torch.meshgrid(
            f.to(dtype=torch.float32),
            torch.arange(0, height, dtype=torch.float32),
            torch.arange(0, width, dtype=torch.float32),
            indexing="ij",
        )
        ct, st = rope_axis(DT, GT, cfg.rope_theta)
        cy, sy = rope_axis(DY, GY, cfg.rope_theta)
        cx, sx = rope_axis(DX, GX, cfg.rope_theta)
        out.append(torch.cat([ct, cy, cx, st, sy, sx], dim=0))
    return torch.stack(out, dim=0)


def apply_rope(x: Tensor, freqs_cis: Tensor) -> Tensor:
    """ref: hunyuan_video_packed.py:80-86.  x [B,S,H,D]; freqs_cis [B,S,2D] fp32 (cos|sin);
    interleaved pairs (2i,2i+1); fp32 math then `.to(x)`."""
    cos, sin = freqs_cis.unsqueeze(-2).chunk(2, dim=-1)
    xr, xi = x.unflatten(-1, (-1, 2)).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + rot.float() * sin).to(x)


def conv3d_patch(P, name: str, x: Tensor) -> Tensor:
    """Conv3d with kernel == stride (ref: hunyuan_video_packed.py:669,675-677) -> [B,tokens,C]."""
    w = P[name + ".weight"]
    y = F.conv3d(x, w, P.get(name + ".bias"), stride=tuple(w.shape[2:]))
    return y


def pack_inputs(P, cfg: DiTConfig, latents, latent_indices=None,
                clean_latents=None, clean_latent_indices=None,
                clean_latents_2x=None, clean_latent_2x_indices=None,
                clean_latents_4x=None, clean_latent_4x_indices=None) -> Tuple[Tensor, Tensor]:
    """ref: hunyuan_video_packed.py:822-879.  Returns (hidden [B,Si,d], rope [B,Si,2*Dh] fp32),
    token order [4x | 2x | 1x clean | noisy]."""
    h = conv3d_patch(P, "x_embedder.proj", latents)
    B, C, T, H, W = h.shape
    if latent_indices is None:
        latent_indices = torch.arange(0, T).unsqueeze(0).expand(B, -1)
    h = h.flatten(2).transpose(1, 2)
    rope = rope_table(cfg, latent_indices, H, W).flatten(2).transpose(1, 2)

    if clean_latents is not None and clean_latent_indices is not None:
        c = conv3d_patch(P, "clean_x_embedder.proj", clean_latents.to(h)).flatten(2).transpose(1, 2)
        r = rope_table(cfg, clean_latent_indices, H, W).flatten(2).transpose(1, 2)
        h = torch.cat([c, h], dim=1)
        rope = torch.cat([r, rope], dim=1)

    if clean_latents_2x is not None and clean_latent_2x_indices is not None:
        c = pad_for_3d_conv(clean_latents_2x.to(h), (2, 4, 4))
        c = conv3d_patch(P, "clean_x_embedder.proj_2x", c).flatten(2).transpose(1, 2)
        r = rope_table(cfg, clean_latent_2x_indices, H, W)
        r = center_down_sample_3d(pad_for_3d_conv(r, (2, 2, 2)), (2, 2, 2)).flatten(2).transpose(1, 2)
        h = torch.cat([c, h], dim=1)
        rope = torch.cat([r, rope], dim=1)

    if clean_latents_4x is not None and clean_latent_4x_indices is not None:
        c = pad_for_3d_conv(clean_latents_4x.to(h), (4, 8, 8))
        c = conv3d_patch(P, "clean_x_embedder.proj_4x", c).flatten(2).transpose(1, 2)
        r = rope_table(cfg, clean_latent_4x_indices, H, W)
        r = center_down_sample_3d(pad_for_3d_conv(r, (4, 4, 4)), (4, 4, 4)).flatten(2).transpose(1, 2)
        h = torch.cat([c, h], dim=1)
        rope = torch.cat([r, rope], dim=1)

    return h, rope


def get_cu_seqlens(text_mask: Tensor, img_len: int) -> Tensor:
    """ref: hunyuan_video_packed.py:63-77 (device='cuda' there; host int32 here).
    Two varlen segments per sample: valid tokens, then the padding tail."""
    B = text_mask.shape[0]
    text_len = text_mask.sum(dim=1)
    max_len = text_mask.shape[1] + img_len
    cu = torch.zeros([2 * B + 1], dtype=torch.int32)
    for i in range(B):
        cu[2 * i + 1] = i * max_len + int(text_len[i]) + img_len
        cu[2 * i + 2] = (i + 1) * max_len
    return cu


# ----------------------------------------------------------------------------
# conditioning embedders
# ----------------------------------------------------------------------------

def time_guidance_text_embed(P, timestep, guidance, pooled) -> Tensor:
    """ref: hunyuan_video_packed.py:200-212."""
    te = mlp_silu(P, "time_text_embed.timestep_embedder", sinusoid_256(timestep).to(pooled.dtype))
    ge = mlp_silu(P, "time_text_embed.guidance_embedder", sinusoid_256(guidance).to(pooled.dtype))
    return (te + ge) + mlp_silu(P, "time_text_embed.text_embedder", pooled)


def refiner_attention(P, pre: str, heads: int, x: Tensor, mask: Optional[Tensor]) -> Tensor:
    """[diffusers, unpinned] Attention + AttnProcessor2_0: q/k/v -> [B,H,S,Dh] -> SDPA with
    the bool mask broadcast over heads -> to_out.0.  Call site ref :286-290."""
    B, S, d = x.shape
    q = linear(P, pre + ".to_q", x).view(B, S, heads, -1).transpose(1, 2)
    k = linear(P, pre + ".to_k", x).view(B, S, heads, -1).transpose(1, 2)
    v = linear(P, pre + ".to_v", x).view(B, S, heads, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    o = o.transpose(1, 2).reshape(B, S, d).to(q.dtype)
    return linear(P, pre + ".to_out.0", o)


def token_refiner(P, cfg: DiTConfig, text: Tensor, timestep: Tensor, mask: Optional[Tensor]) -> Tensor:
    """ref: hunyuan_video_packed.py:376-394 (+ :326-345 mask, :278-298 block, :242-248 gates)."""
    if mask is None:
        pooled = text.mean(dim=1)
    else:
        mf = mask.float().unsqueeze(-1)
        pooled = ((text * mf).sum(dim=1) / mf.sum(dim=1)).to(text.dtype)
    pre = "context_embedder."
    temb = mlp_silu(P, pre + "time_text_embed.timestep_embedder", sinusoid_256(timestep).to(pooled.dtype)) \
        + mlp_silu(P, pre + "time_text_embed.text_embedder", pooled)
    x = linear(P, pre + "proj_in", text)

    sa_mask = None
    if mask is not None:
        B, L = mask.shape
        m1 = mask.bool().view(B, 1, 1, L).repeat(1, 1, L, 1)
        sa_mask = (m1 & m1.transpose(2, 3)).bool()
        sa_mask[:, :, :, 0] = True                      # ref :340

    for i in range(cfg.num_refiner_layers):
        b = f"{pre}token_refiner.refiner_blocks.{i}"
        n1 = layer_norm(x, P[b + ".norm1.weight"], P[b + ".norm1.bias"])
        a = refiner_attention(P, b + ".attn", cfg.num_attention_heads, n1, sa_mask)
        g = linear(P, b + ".norm_out.linear", F.silu(temb))
        gate_msa, gate_mlp = (t.unsqueeze(1) for t in g.chunk(2, dim=-1))
        x = x + a * gate_msa
        n2 = layer_norm(x, P[b + ".norm2.weight"], P[b + ".norm2.bias"])
        ff = linear(P, b + ".ff.net.2", F.silu(linear(P, b + ".ff.net.0.proj", n2)))   # "linear-silu"
        x = x + ff * gate_mlp
    return x


def image_projection(P, x: Tensor) -> Tensor:
    """ref: hunyuan_video_packed.py:661-663."""
    return linear(P, "image_projection.down", F.silu(linear(P, "image_projection.up", x)))


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------

def ada_mod(P, name: str, temb: Tensor, n: int):
    """emb = Linear(SiLU(temb)).chunk(n) with a token axis.  ref: :450-452, :473-475, :499-501."""
    return linear(P, name, F.silu(temb.unsqueeze(-2))).chunk(n, dim=-1)


def feed_forward_gelu(P, pre: str, x: Tensor) -> Tensor:
    """[diffusers, unpinned] FeedForward('gelu-approximate'): net.0.proj -> tanh-GELU -> net.2."""
    return linear(P, pre + ".net.2", F.gelu(linear(P, pre + ".net.0.proj", x), approximate="tanh"))


def heads_of(x: Tensor, H: int) -> Tensor:
    return x.unflatten(2, (H, -1))


def double_block(P, cfg, i: int, h: Tensor, e: Tensor, temb: Tensor, rope: Tensor, attn_fn=sdpa_nhd):
    """ref: hunyuan_video_packed.py:615-652 (block) and :115-158 (processor)."""
    b = f"transformer_blocks.{i}"
    H = cfg.num_attention_heads
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = ada_mod(P, b + ".norm1.linear", temb, 6)
    csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = ada_mod(P, b + ".norm1_context.linear", temb, 6)
    nh = layer_norm(h) * (1 + sc_a) + sh_a
    ne = layer_norm(e) * (1 + csc_a) + csh_a

    a = b + ".attn"
    q = apply_rope(rms_norm(heads_of(linear(P, a + ".to_q", nh), H), P[a + ".norm_q.weight"]), rope)
    k = apply_rope(rms_norm(heads_of(linear(P, a + ".to_k", nh), H), P[a + ".norm_k.weight"]), rope)
    v = heads_of(linear(P, a + ".to_v", nh), H)
    eq = rms_norm(heads_of(linear(P, a + ".add_q_proj", ne), H), P[a + ".norm_added_q.weight"])
    ek = rms_norm(heads_of(linear(P, a + ".add_k_proj", ne), H), P[a + ".norm_added_k.weight"])
    ev = heads_of(linear(P, a + ".add_v_proj", ne), H)
    o = attn_fn(torch.cat([q, eq], 1), torch.cat([k, ek], 1), torch.cat([v, ev], 1)).flatten(-2)
    L = e.shape[1]
    ao = linear(P, a + ".to_out.0", o[:, :-L])
    co = linear(P, a + ".to_add_out", o[:, -L:])

    h = h + ao * g_a
    e = e + co * cg_a
    nh = layer_norm(h) * (1 + sc_m) + sh_m
    ne = layer_norm(e) * (1 + csc_m) + csh_m
    h = h + g_m * feed_forward_gelu(P, b + ".ff", nh)
    e = e + cg_m * feed_forward_gelu(P, b + ".ff_context", ne)
    return h, e


def single_block(P, cfg, i: int, h: Tensor, e: Tensor, temb: Tensor, rope: Tensor, attn_fn=sdpa_nhd):
    """ref: hunyuan_video_packed.py:537-577 (block) and :161-188 (processor)."""
    b = f"single_transformer_blocks.{i}"
    H = cfg.num_attention_heads
    L = e.shape[1]
    x = torch.cat([h, e], dim=1)
    shift, scale, gate = ada_mod(P, b + ".norm.linear", temb, 3)
    n = layer_norm(x) * (1 + scale) + shift
    mlp = F.gelu(linear(P, b + ".proj_mlp", n), approximate="tanh")

    a = b + ".attn"
    q = rms_norm(heads_of(linear(P, a + ".to_q", n), H), P[a + ".norm_q.weight"])
    k = rms_norm(heads_of(linear(P, a + ".to_k", n), H), P[a + ".norm_k.weight"])
    v = heads_of(linear(P, a + ".to_v", n), H)
    q = torch.cat([apply_rope(q[:, :-L], rope), q[:, -L:]], dim=1)
    k = torch.cat([apply_rope(k[:, :-L], rope), k[:, -L:]], dim=1)
    o = attn_fn(q, k, v).flatten(-2)

    x = gate * linear(P, b + ".proj_out", torch.cat([o, mlp], dim=2)) + x
    return x[:, :-L], x[:, -L:]


# ----------------------------------------------------------------------------
# TeaCache state + full forward
# ----------------------------------------------------------------------------

TEACACHE_POLY = np.poly1d([7.33226126e+02, -4.01131952e+02, 6.75869174e+01, -3.14987800e+00, 9.61237896e-02])


@dataclass
class TeaCacheState:
    """ref: hunyuan_video_packed.py:785-813."""
    enable: bool = False
    num_steps: int = 25
    rel_l1_thresh: float = 0.15
    cnt: int = 0
    acc: float = 0.0
    prev_mod: Optional[Tensor] = None
    prev_residual: Optional[Tensor] = None
    hits: int = 0
    misses: int = 0
    queries: int = 0
    trace: list = field(default_factory=list)
    # dtype the statistic is taken in.  None = the dtype of the run (the reference's behaviour: a bf16 run subtracts,
    # averages and divides in bf16, ref :957-958).  An fp32 run that checks a bf16 product path sets torch.bfloat16.
    stat_dtype: Optional[torch.dtype] = None


def dit_forward(P, cfg: DiTConfig, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask,
                pooled_projections, guidance, latent_indices=None,
                clean_latents=None, clean_latent_indices=None,
                clean_latents_2x=None, clean_latent_2x_indices=None,
                clean_latents_4x=None, clean_latent_4x_indices=None,
                image_embeddings=None, teacache: Optional[TeaCacheState] = None,
                fp32_output: bool = False, taps: Optional[dict] = None, attn_fn=sdpa_nhd) -> Tensor:
    """ref: hunyuan_video_packed.py:881-1062, B==1 branch (:927-932) and B>1 via per-sample crop
    (mathematically what the varlen segments compute for the valid tokens)."""
    B, C, T, Hh, Ww = hidden_states.shape
    p, pt = cfg.patch_size, cfg.patch_size_t
    t2, h2, w2 = T // pt, Hh // p, Ww // p
    ctx_len = t2 * h2 * w2

    h, rope = pack_inputs(P, cfg, hidden_states, latent_indices, clean_latents, clean_latent_indices,
                          clean_latents_2x, clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices)
    temb = time_guidance_text_embed(P, timestep, guidance, pooled_projections)
    e = token_refiner(P, cfg, encoder_hidden_states, timestep, encoder_attention_mask)
    mask = encoder_attention_mask
    if cfg.has_image_proj:
        assert image_embeddings is not None, "You must use image embeddings!"
        extra = image_projection(P, image_embeddings)
        e = torch.cat([extra, e], dim=1)                 # image tokens BEFORE text (ref :922-923)
        mask = torch.cat([torch.ones((B, extra.shape[1]), dtype=mask.dtype), mask], dim=1)
    if taps is not None:
        taps["packed_hidden"], taps["rope"], taps["temb"], taps["text_ctx"] = h, rope, temb, e

    assert B == 1, "oracle restates the B==1 crop branch (ref :927-932)"
    text_len = int(mask.sum().item())
    e = e[:, :text_len]

    def run_blocks(h, e):
        for i in range(cfg.num_layers):
            h, e = double_block(P, cfg, i, h, e, temb, rope, attn_fn)
            if taps is not None:
                taps[f"double.{i}"] = (h, e)
        for i in range(cfg.num_single_layers):
            h, e = single_block(P, cfg, i, h, e, temb, rope, attn_fn)
            if taps is not None:
                taps[f"single.{i}"] = (h, e)
        return h, e

    tc = teacache
    if tc is not None and tc.enable:
        sh, sc = ada_mod(P, "transformer_blocks.0.norm1.linear", temb, 6)[:2]
        mod = layer_norm(h) * (1 + sc) + sh              # ref :945
        tc.queries += 1
        rel = 0.0
        if tc.cnt == 0 or tc.cnt == tc.num_steps - 1:
            calc = True
            tc.acc = 0.0
            tc.misses += 1
        else:
            cur_s, prev_s = (mod, tc.prev_mod) if tc.stat_dtype is None else (mod.to(tc.stat_dtype), tc.prev_mod.to(tc.stat_dtype))
            rel = ((cur_s - prev_s).abs().mean() / (prev_s.abs().mean() + 1e-8)).item()
            tc.acc += float(TEACACHE_POLY(rel))
            calc = tc.acc >= tc.rel_l1_thresh
            if calc:
                tc.acc = 0.0
                tc.misses += 1
            else:
                tc.hits += 1
        tc.prev_mod = mod
        tc.cnt = (tc.cnt + 1) % tc.num_steps
        tc.trace.append((calc, rel))
        if not calc:
            h = h + tc.prev_residual
        else:
            h0 = h.clone()
            h, e = run_blocks(h, e)
            tc.prev_residual = h - h0
    else:
        h, e = run_blocks(h, e)

    scale, shift = ada_mod(P, "norm_out.linear", temb, 2)      # (scale, shift) order: ref :501
    h = layer_norm(h) * (1 + scale) + shift
    h = h[:, -ctx_len:, :]
    if fp32_output:                                           # ref :1048-1051
        h = h.float()
        out = F.linear(h, P["proj_out.weight"].float(), P["proj_out.bias"].float())
    else:
        out = linear(P, "proj_out", h)
    if taps is not None:
        taps["head"] = out
    # 'b (t h w) (c pt ph pw) -> b c (t pt) (h ph) (w pw)'   ref :1055-1057
    co = cfg.out_channels
    out = out.view(B, t2, h2, w2, co, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7)
    return out.reshape(B, co, t2 * pt, h2 * p, w2 * p)


# ----------------------------------------------------------------------------
# synthetic parameters (shared by tests, goldens and bench: same seed -> same weights)
# ----------------------------------------------------------------------------

def param_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict names and shapes of the packed DiT (names: SURVEY.md 8(b))."""
    d = cfg.inner_dim
    Dh = cfg.attention_head_dim
    m = int(d * cfg.mlp_ratio)
    S: Dict[str, Tuple[int, ...]] = {}

    def lin(name, i, o):
        S[name + ".weight"] = (o, i)
        S[name + ".bias"] = (o,)

    def conv(name, ci, co, k):
        S[name + ".weight"] = (co, ci) + tuple(k)
        S[name + ".bias"] = (co,)

    conv("x_embedder.proj", cfg.in_channels, d, (cfg.patch_size_t, cfg.patch_size, cfg.patch_size))
    ce = "context_embedder."
    lin(ce + "time_text_embed.timestep_embedder.linear_1", 256, d)
    lin(ce + "time_text_embed.timestep_embedder.linear_2", d, d)
    lin(ce + "time_text_embed.text_embedder.linear_1", cfg.text_embed_dim, d)
    lin(ce + "time_text_embed.text_embedder.linear_2", d, d)
    lin(ce + "proj_in", cfg.text_embed_dim, d)
    for i in range(cfg.num_refiner_layers):
        b = f"{ce}token_refiner.refiner_blocks.{i}"
        S[b + ".norm1.weight"] = (d,); S[b + ".norm1.bias"] = (d,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{b}.attn.{n}", d, d)
        S[b + ".norm2.weight"] = (d,); S[b + ".norm2.bias"] = (d,)
        lin(b + ".ff.net.0.proj", d, m)
        lin(b + ".ff.net.2", m, d)
        lin(b + ".norm_out.linear", d, 2 * d)
    for n in ("timestep_embedder", "guidance_embedder"):
        lin(f"time_text_embed.{n}.linear_1", 256, d)
        lin(f"time_text_embed.{n}.linear_2", d, d)
    lin("time_text_embed.text_embedder.linear_1", cfg.pooled_projection_dim, d)
    lin("time_text_embed.text_embedder.linear_2", d, d)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        lin(b + ".norm1.linear", d, 6 * d)
        lin(b + ".norm1_context.linear", d, 6 * d)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(f"{b}.attn.{n}", d, d)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            S[f"{b}.attn.{n}.weight"] = (Dh,)
        for f_ in ("ff", "ff_context"):
            lin(f"{b}.{f_}.net.0.proj", d, m)
            lin(f"{b}.{f_}.net.2", m, d)
    for i in range(cfg.num_single_layers):
        b = f"single_transformer_blocks.{i}"
        lin(b + ".norm.linear", d, 3 * d)
        for n in ("to_q", "to_k", "to_v"):
            lin(f"{b}.attn.{n}", d, d)
        for n in ("norm_q", "norm_k"):
            S[f"{b}.attn.{n}.weight"] = (Dh,)
        lin(b + ".proj_mlp", d, m)
        lin(b + ".proj_out", d + m, d)
    lin("norm_out.linear", d, 2 * d)
    lin("proj_out", d, cfg.patch_size_t * cfg.patch_size * cfg.patch_size * cfg.out_channels)
    if cfg.has_image_proj:
        lin("image_projection.up", cfg.image_proj_dim, 3 * d)
        lin("image_projection.down", 3 * d, d)
    if cfg.has_clean_x_embedder:
        conv("clean_x_embedder.proj", 16, d, (1, 2, 2))
        conv("clean_x_embedder.proj_2x", 16, d, (2, 4, 4))
        conv("clean_x_embedder.proj_4x", 16, d, (4, 8, 8))
    return S


def synth_params(cfg: DiTConfig, seed: int = 0, dtype=torch.float32, std: float = 0.02,
                 bias_std: float = 0.02) -> Dict[str, Tensor]:
    """Seeded synthetic weights, drawn per tensor in sorted-name order from a CPU generator:
    weights N(0, fan-in-scaled), biases N(0, bias_std^2), norm weights 1 + N(0, 0.1^2).
    Matrix weights use std = 1/sqrt(fan_in) so activations stay O(1) through the blocks."""
    g = torch.Generator("cpu").manual_seed(seed)
    out = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * bias_std
        elif len(shape) == 1:                             # norm weights
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        out[name] = t.to(dtype)
    return out
