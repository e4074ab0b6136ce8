"""HunyuanVideoTransformer3DModelPacked on MI355X: host mirror of the reference's module interface.

Same class name, constructor arguments, forward signature, state-dict key names, TeaCache attributes
and error behaviour as ref diffusers_helper/models/hunyuan_video_packed.py:698-1062, so
diffusers_helper/k_diffusion/wrapper.py:43 and framepack/worker.py drive it unchanged.  The arithmetic
of the 60 transformer blocks runs in libframepack_hip.so (hand-written gfx950 kernels, C ABI in
include/framepack_hip.h); torch provides device memory, the stream and the tiny step-level embedders.

Data layout in HBM (B == 1, S = Si + St tokens, d = heads*128):
  X   [S, d]      residual stream, image tokens first then [image-proj | text] tokens: the double-stream
                  blocks use the two row ranges, the single-stream blocks the whole buffer (no cat/split)
  N   [S, d]      LayerNorm+modulate output
  Y   [S, 7d]     projection workspace: double blocks use [q|k|v] (3d) and the FF hidden (4d);
                  single blocks the fused [k|v|q|mlp]; attention writes O over q, so the output
                  projection reads [o] or [o|mlp] in place (no cat of ref :144-146, :165, :569)
There is no CPU path: forward() raises if the inputs are not on a GPU or the library is missing.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib
from . import ops
from .packing import RopeCache, get_cu_seqlens, pad_to_multiple, patch_rows, unpatchify

BF16 = torch.bfloat16


# --------------------------------------------------------------------------------------------
# parameter containers (names = the reference's / diffusers' state-dict names)
# --------------------------------------------------------------------------------------------
class _NormWeight(nn.Module):
    """RMSNorm parameter holder (`norm_q.weight` ...); applied inside the fused K2 kernel."""

    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class _MLP2(nn.Module):
    """linear_2(silu(linear_1(x))): TimestepEmbedding / PixArtAlphaTextProjection(act_fn='silu') (ref :191-231).  The
    input is one vector per sample, so both linears are weight-streaming GEMVs: K7 (`fp_silu_gemv_bf16`), the first
    without the SiLU, the second with it - the same rounding points as the reference's bf16 modules (linear_1's output
    and the SiLU's are bf16)."""

    def __init__(self, d_in: int, d: int):
        super().__init__()
        self.linear_1 = nn.Linear(d_in, d)
        self.linear_2 = nn.Linear(d, d)

    def forward(self, x):
        l1, l2 = self.linear_1, self.linear_2
        rows = []
        for r in x.reshape(-1, x.shape[-1]):
            h = ops.silu_gemv(r, l1.weight.detach(), l1.bias.detach(), apply_silu=False)
            rows.append(ops.silu_gemv(h, l2.weight.detach(), l2.bias.detach(), apply_silu=True))
        return torch.stack(rows).reshape(x.shape[:-1] + (l2.out_features,))


class _ActProj(nn.Module):
    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.proj = nn.Linear(d_in, d_out)


class _FeedForward(nn.Module):
    """Parameter layout of diffusers FeedForward: net.0.proj, net.2 (net.1 is the dropout slot)."""

    def __init__(self, d: int, mult: float):
        super().__init__()
        inner = int(d * mult)
        self.net = nn.ModuleList([_ActProj(d, inner), nn.Identity(), nn.Linear(inner, d)])


class _Attention(nn.Module):
    """Parameter layout of diffusers Attention as the reference constructs it (ref :265-271, 519-530, 595-607)."""

    def __init__(self, d: int, heads: int, head_dim: int, qk_norm: bool, added_kv: bool, pre_only: bool, eps=1e-6):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        if qk_norm:
            self.norm_q, self.norm_k = _NormWeight(head_dim, eps), _NormWeight(head_dim, eps)
        if added_kv:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
            if qk_norm:
                self.norm_added_q, self.norm_added_k = _NormWeight(head_dim, eps), _NormWeight(head_dim, eps)
            self.to_add_out = nn.Linear(d, d)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(d, d), nn.Identity()])


class _AdaLinear(nn.Module):
    """`<name>.linear`: emb = Linear(SiLU(temb)) (ref :435-503, :234-248); evaluated by the K7 GEMV."""

    def __init__(self, d: int, n_out: int):
        super().__init__()
        self.linear = nn.Linear(d, n_out)


class HunyuanVideoTransformerBlock(nn.Module):          # ref :580-652
    def __init__(self, heads: int, head_dim: int, mlp_ratio: float, qk_norm: str = "rms_norm"):
        super().__init__()
        d = heads * head_dim
        self.norm1 = _AdaLinear(d, 6 * d)
        self.norm1_context = _AdaLinear(d, 6 * d)
        self.attn = _Attention(d, heads, head_dim, qk_norm == "rms_norm", added_kv=True, pre_only=False)
        self.ff = _FeedForward(d, mlp_ratio)
        self.ff_context = _FeedForward(d, mlp_ratio)


class HunyuanVideoSingleTransformerBlock(nn.Module):    # ref :506-577
    def __init__(self, heads: int, head_dim: int, mlp_ratio: float = 4.0, qk_norm: str = "rms_norm"):
        super().__init__()
        d = heads * head_dim
        self.attn = _Attention(d, heads, head_dim, qk_norm == "rms_norm", added_kv=False, pre_only=True)
        self.norm = _AdaLinear(d, 3 * d)
        self.proj_mlp = nn.Linear(d, int(d * mlp_ratio))
        self.proj_out = nn.Linear(d + int(d * mlp_ratio), d)


class _RefinerBlock(nn.Module):                         # ref :251-298
    def __init__(self, heads: int, head_dim: int, mlp_ratio: float):
        super().__init__()
        d = heads * head_dim
        self.norm1 = nn.LayerNorm(d, eps=1e-6)
        self.attn = _Attention(d, heads, head_dim, qk_norm=False, added_kv=False, pre_only=False)
        self.norm2 = nn.LayerNorm(d, eps=1e-6)
        self.ff = _FeedForward(d, mlp_ratio)
        self.norm_out = _AdaLinear(d, 2 * d)


class _TokenRefinerStack(nn.Module):
    def __init__(self, heads, head_dim, n, mlp_ratio):
        super().__init__()
        self.refiner_blocks = nn.ModuleList([_RefinerBlock(heads, head_dim, mlp_ratio) for _ in range(n)])


class _TimeTextEmbed(nn.Module):
    def __init__(self, d, pooled_dim, guidance: bool):
        super().__init__()
        self.timestep_embedder = _MLP2(256, d)
        if guidance:
            self.guidance_embedder = _MLP2(256, d)
        self.text_embedder = _MLP2(pooled_dim, d)


def _sinusoid_256(t: torch.Tensor) -> torch.Tensor:
    """Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): fp32 [cos | sin] (ref :195, 219)."""
    f = torch.exp(-np.log(10000.0) * torch.arange(128, dtype=torch.float32, device=t.device) / 128)
    ang = t[:, None].float() * f[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _bf16_round(x: float) -> float:
    """Round a host float to the nearest bf16 value (what a 0-dim bf16 tensor would hold)."""
    return float(torch.tensor(x, dtype=torch.float32).to(BF16))


class _MaskInfo:
    """Step-invariant facts about an encoder_attention_mask, read back from the device ONCE per mask tensor (the
    reference syncs on `mask.sum().item()` every step, ref :930): number of valid tokens, whether they form a prefix,
    and the rows the refiner's attention runs over (valid tokens plus token 0, ref :337-340)."""

    def __init__(self, mask: Optional[torch.Tensor]):
        if mask is None:
            self.n_valid, self.prefix, self.rows, self.invalid = None, True, None, None
            return
        m = mask.reshape(-1).bool()
        host = m.cpu()
        self.n_valid = int(host.sum())
        self.prefix = bool(host[:self.n_valid].all())
        att = host.clone()
        att[0] = True                                                   # column 0 is visible to every row
        self.n_att = int(att.sum())
        self.att_prefix = bool(att[:self.n_att].all())
        self.rows = None if self.att_prefix else att.nonzero().flatten().to(mask.device)
        self.invalid = None if self.prefix and self.n_valid == m.numel() else (~host).nonzero().flatten().to(mask.device)


class HunyuanVideoTokenRefiner(nn.Module):              # ref :348-394
    """Text-context embedder: 512 padded tokens, 2 small blocks, once per step (< 0.3 % of the FLOPs).  Its linears
    run on K5 (`fp_gemm_bf16`, gated residuals in the epilogue), its masked attention on K1: under the reference's
    mask (outer AND of the key mask, column 0 forced visible, ref :337-340) a valid token attends to the valid tokens
    plus token 0, and a padding token to token 0 alone - i.e. K1 over those rows, and v[0] for the padding rows.
    The two affine LayerNorms per block are K3's affine form, the two vector MLPs and `norm_out.linear` K7 GEMVs; what
    stays on torch are elementwise ops (the masked mean of the text states, the FF's SiLU)."""

    def __init__(self, in_channels, heads, head_dim, num_layers, mlp_ratio=4.0):
        super().__init__()
        d = heads * head_dim
        self.heads = heads
        self.time_text_embed = _TimeTextEmbed(d, in_channels, guidance=False)
        self.proj_in = nn.Linear(in_channels, d)
        self.token_refiner = _TokenRefinerStack(heads, head_dim, num_layers, mlp_ratio)
        self._packed = None            # fused q|k|v weights per block (rebuilt after .to() / load_state_dict)

    def _apply(self, fn, *a, **k):                      # .to() / .cuda(): the fused copies go stale
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _pack(self):
        with torch.inference_mode(False), torch.no_grad():
            P = []
            for blk in self.token_refiner.refiner_blocks:
                a = blk.attn
                e = SimpleNamespace()
                e.w_qkv = torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], dim=0).detach().to(BF16).contiguous()
                e.b_qkv = torch.cat([a.to_q.bias, a.to_k.bias, a.to_v.bias], dim=0).detach().to(BF16).contiguous()
                P.append(e)
            self._packed = P
        return P

    def forward(self, text, timestep, mask, info: Optional[_MaskInfo] = None):
        if not text.is_cuda:
            raise RuntimeError("HunyuanVideoTokenRefiner (MI355X build) runs on the GPU only; there is no CPU fallback")
        with torch.cuda.device(text.device):
            return self._forward(text, timestep, mask, info)

    def _forward(self, text, timestep, mask, info):
        B, L, _ = text.shape
        assert B == 1, "the packed forward feeds the refiner one sample at a time"
        info = info or _MaskInfo(mask)
        if mask is None:
            pooled = text.mean(dim=1)
        else:
            mf = mask.float().unsqueeze(-1)
            pooled = ((text * mf).sum(dim=1) / mf.sum(dim=1)).to(text.dtype)
        temb = self.time_text_embed.timestep_embedder(_sinusoid_256(timestep).to(pooled.dtype)) \
            + self.time_text_embed.text_embedder(pooled)
        P = self._packed or self._pack()
        H = self.heads
        d = self.proj_in.out_features
        x = ops.gemm(text[0], self.proj_in.weight.detach(), self.proj_in.bias.detach())          # [L, d]
        for blk, e in zip(self.token_refiner.refiner_blocks, P):
            gates = ops.silu_gemv(temb[0], blk.norm_out.linear.weight.detach(), blk.norm_out.linear.bias.detach())
            g_msa, g_mlp = gates[:d], gates[d:]
            n1 = ops.layernorm_affine(x, blk.norm1.weight.detach(), blk.norm1.bias.detach(), blk.norm1.eps)
            qkv = ops.gemm(n1, e.w_qkv, e.b_qkv)                                                   # [L, q|k|v]
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            if mask is None:
                att = ops.attention(q, k, v, H)
            else:
                att = v[0:1].expand(L, d).contiguous()            # a padding row sees token 0 only: softmax = 1
                if info.att_prefix:
                    n = info.n_att
                    ops.attention(q[:n], k[:n], v[:n], H, out=att[:n])
                else:
                    sub = qkv.index_select(0, info.rows)
                    att.index_copy_(0, info.rows, ops.attention(sub[:, :d], sub[:, d:2 * d], sub[:, 2 * d:], H))
                if info.invalid is not None and not info.prefix:  # an invalid token 0 took part as a query above
                    att.index_copy_(0, info.invalid, v[0:1].expand(info.invalid.numel(), d))
            ops.gemm(att, blk.attn.to_out[0].weight.detach(), blk.attn.to_out[0].bias.detach(), out=x,
                     epilogue=ops.EPI_GATE_RESIDUAL, resid=x, gate=g_msa)
            n2 = ops.layernorm_affine(x, blk.norm2.weight.detach(), blk.norm2.bias.detach(), blk.norm2.eps, out=n1)
            hdn = F.silu(ops.gemm(n2, blk.ff.net[0].proj.weight.detach(), blk.ff.net[0].proj.bias.detach()))
            ops.gemm(hdn, blk.ff.net[2].weight.detach(), blk.ff.net[2].bias.detach(), out=x,
                     epilogue=ops.EPI_GATE_RESIDUAL, resid=x, gate=g_mlp)
        return x.unsqueeze(0)


class ClipVisionProjection(nn.Module):                  # ref :655-663
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = nn.Linear(in_channels, out_channels * 3)
        self.down = nn.Linear(out_channels * 3, out_channels)

    def forward(self, x):
        """down(silu(up(x))) on K5 (once per section: the result is cached across steps, model._cached)."""
        rows = x.reshape(-1, x.shape[-1])
        h = F.silu(ops.gemm(rows, self.up.weight.detach(), self.up.bias.detach()))
        return ops.gemm(h, self.down.weight.detach(), self.down.bias.detach()).reshape(x.shape[:-1] + (self.down.out_features,))


class HunyuanVideoPatchEmbed(nn.Module):                # ref :666-669
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class HunyuanVideoPatchEmbedForCleanLatents(nn.Module):  # ref :672-677
    def __init__(self, inner_dim):
        super().__init__()
        self.proj = nn.Conv3d(16, inner_dim, kernel_size=(1, 2, 2), stride=(1, 2, 2))
        self.proj_2x = nn.Conv3d(16, inner_dim, kernel_size=(2, 4, 4), stride=(2, 4, 4))
        self.proj_4x = nn.Conv3d(16, inner_dim, kernel_size=(4, 8, 8), stride=(4, 8, 8))


class Transformer2DModelOutput(SimpleNamespace):
    """`return_dict=True` result (the reference returns diffusers' dataclass with a `.sample` field)."""


# --------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------
class HunyuanVideoTransformer3DModelPacked(nn.Module):
    """ref diffusers_helper/models/hunyuan_video_packed.py:698-774 (constructor and attributes)."""

    def __init__(self, in_channels: int = 16, out_channels: int = 16, num_attention_heads: int = 24,
                 attention_head_dim: int = 128, num_layers: int = 20, num_single_layers: int = 40,
                 num_refiner_layers: int = 2, mlp_ratio: float = 4.0, patch_size: int = 2, patch_size_t: int = 1,
                 qk_norm: str = "rms_norm", guidance_embeds: bool = True, text_embed_dim: int = 4096,
                 pooled_projection_dim: int = 768, rope_theta: float = 256.0, rope_axes_dim=(16, 56, 56),
                 has_image_proj=False, image_proj_dim=1152, has_clean_x_embedder=False) -> None:
        super().__init__()
        if attention_head_dim != 128:
            raise ValueError("the gfx950 kernels are built for attention_head_dim == 128")
        self.config = dict(in_channels=in_channels, out_channels=out_channels, num_attention_heads=num_attention_heads,
                           attention_head_dim=attention_head_dim, num_layers=num_layers,
                           num_single_layers=num_single_layers, num_refiner_layers=num_refiner_layers,
                           mlp_ratio=mlp_ratio, patch_size=patch_size, patch_size_t=patch_size_t, qk_norm=qk_norm,
                           guidance_embeds=guidance_embeds, text_embed_dim=text_embed_dim,
                           pooled_projection_dim=pooled_projection_dim, rope_theta=rope_theta,
                           rope_axes_dim=tuple(rope_axes_dim), has_image_proj=has_image_proj,
                           image_proj_dim=image_proj_dim, has_clean_x_embedder=has_clean_x_embedder)
        inner_dim = num_attention_heads * attention_head_dim
        out_channels = out_channels or in_channels
        self.heads, self.head_dim = num_attention_heads, attention_head_dim

        self.x_embedder = HunyuanVideoPatchEmbed((patch_size_t, patch_size, patch_size), in_channels, inner_dim)
        self.context_embedder = HunyuanVideoTokenRefiner(text_embed_dim, num_attention_heads, attention_head_dim,
                                                         num_layers=num_refiner_layers)
        self.time_text_embed = _TimeTextEmbed(inner_dim, pooled_projection_dim, guidance=True)
        self.clean_x_embedder = None
        self.image_projection = None
        self.transformer_blocks = nn.ModuleList([
            HunyuanVideoTransformerBlock(num_attention_heads, attention_head_dim, mlp_ratio, qk_norm)
            for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList([
            HunyuanVideoSingleTransformerBlock(num_attention_heads, attention_head_dim, mlp_ratio, qk_norm)
            for _ in range(num_single_layers)])
        self.norm_out = _AdaLinear(inner_dim, 2 * inner_dim)
        self.proj_out = nn.Linear(inner_dim, patch_size_t * patch_size * patch_size * out_channels)

        self.inner_dim = inner_dim
        self.use_gradient_checkpointing = False
        self.enable_teacache = False
        if has_image_proj:
            self.install_image_projection(image_proj_dim)
        if has_clean_x_embedder:
            self.install_clean_x_embedder()
        self.high_quality_fp32_output_for_inference = False

        self._rope_cache = RopeCache(rope_axes_dim, rope_theta)
        self._packed = None            # fused weight buffers, built lazily (see _pack)
        self._invariant = {}           # step-invariant results keyed by the identity of their inputs (see _cached)
        self._ws = {}                  # grow-only activation backing store per device (see _workspace)
        self.taps = None               # optional dict: per-block outputs for parity tests
        self.sp_group, self.sp_world, self.sp_rank = None, 1, 0    # Ulysses sequence parallelism (off)
        self.sp_on = False             # the sequence-parallel code path (a world of one rank can be made to take it)
        self.sp_chunks = None          # head chunks per exchange (None: ulysses.pick_chunks decides)
        self._sp_bufs = {}             # the exchange's buffers, allocated once per shape (ulysses.exchange_buffers)
        self._sp_total = 0
        # single blocks under sequence parallelism: the MLP half of the in-projection runs while q | k | v are exchanged
        # (_single_block); FRAMEPACK_SP_OVERLAP_MLP=0: one fused in-projection, exchange after it
        self.sp_overlap_mlp = os.environ.get("FRAMEPACK_SP_OVERLAP_MLP", "1") != "0"
        self.fp8_compute = False       # block linears on the fp8 MFMA (enable_fp8_compute)
        # text stream of the double blocks on a low-priority side stream (_double_block); FRAMEPACK_TEXT_OVERLAP=0: off
        self.text_stream_overlap = os.environ.get("FRAMEPACK_TEXT_OVERLAP", "1") != "0"
        self._side_streams = {}
        # double blocks: the image and the text linear of a pair in ONE launch over the rows of the shared buffer
        # (ops.gemm_rows2; _double_block); FRAMEPACK_GROUP_TEXT=0: separate launches (the text ones on the side stream)
        self.group_text_gemms = os.environ.get("FRAMEPACK_GROUP_TEXT", "1") != "0"
        # K2 (RMSNorm + RoPE of q, k) in the epilogue of the GEMM that produces them (_ln_lin_qk): bit-identical, but
        # measured 0.8 % SLOWER per step than the separate pass, so off unless FRAMEPACK_FUSE_QK=1
        self.fuse_qk_norm = os.environ.get("FRAMEPACK_FUSE_QK", "0") == "1"
        self.fp8_attention = False     # joint attention with e4m3 Q, K, P, V (enable_fp8_attention)

    # ---- reference API ---------------------------------------------------------------------
    def install_image_projection(self, in_channels):
        self.image_projection = ClipVisionProjection(in_channels=in_channels, out_channels=self.inner_dim)
        self.config["has_image_proj"] = True
        self.config["image_proj_dim"] = in_channels

    def install_clean_x_embedder(self):
        self.clean_x_embedder = HunyuanVideoPatchEmbedForCleanLatents(self.inner_dim)
        self.config["has_clean_x_embedder"] = True

    def initialize_teacache(self, enable_teacache=True, num_steps=25, rel_l1_thresh=0.15, _=None):
        """ref :785-813."""
        self.enable_teacache = enable_teacache
        self.cnt = 0
        self.num_steps = num_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.teacache_rescale_func = np.poly1d(
            [7.33226126e+02, -4.01131952e+02, 6.75869174e+01, -3.14987800e+00, 9.61237896e-02])
        self.cache_hits = 0
        self.cache_misses = 0
        self.cache_queries = 0

    def enable_sequence_parallel(self, group=None, force: bool = False):
        """Shard the image tokens over the ranks of `group` (default: the world) and run the joint attention
        Ulysses-style (framepack_amd/ulysses.py).  Every rank calls forward() with the SAME inputs and gets
        the SAME full output; weights are replicated.  No reference counterpart.
        A group of ONE rank keeps the single-GPU path unless `force` (the RCCL test on one GPU: every exchange,
        gather and all-reduce of the sequence-parallel path then runs through the communicator, on one rank)."""
        import torch.distributed as dist
        self.sp_group = group
        self.sp_world = dist.get_world_size(group)
        self.sp_rank = dist.get_rank(group)
        if self.heads % self.sp_world != 0:
            raise ValueError(f"{self.heads} heads do not divide over {self.sp_world} ranks")
        self.sp_on = self.sp_world > 1 or force
        self._ws = {}
        self._sp_bufs = {}

    def _cached(self, slot: str, src: torch.Tensor, fn):
        """Cross-step cache of work that depends only on a tensor the caller passes unchanged for every step of a
        section: the clean-latent patch embeddings and the image projection, which the
        reference recomputes 25 times (`ref :841-866, :917-924`; the clean latents even arrive as fp32 CPU tensors and
        are re-uploaded every step).  The key is the tensor's identity (storage address, shape, dtype, in-place
        version counter): a different or modified tensor recomputes.  Tensors made under torch.inference_mode (how the
        reference's worker runs) carry no version counter; for those a CPU tensor is fingerprinted by a strided sample
        of its values, a GPU tensor by identity alone (the worker builds fresh tensors per section and never writes
        into them).  One entry per slot."""
        if not src.is_inference():
            stamp = src._version
        elif src.device.type == "cpu":
            flat = src.reshape(-1)                        # small tensors (masks, indices) whole, large ones sampled
            stamp = tuple(flat[:: 1 if flat.numel() <= 4096 else flat.numel() // 61].double().tolist())
        else:
            stamp = None
        key = (src.data_ptr(), tuple(src.shape), src.dtype, str(src.device), stamp)
        hit = self._invariant.get(slot)
        if hit is not None and hit[0] == key and hit[1] is src:
            return hit[2]
        val = fn()
        self._invariant[slot] = (key, src, val)       # holding `src` keeps its address from being reused
        return val

    def invalidate_packed(self):
        """Drop the fused weight buffers (rebuilt on the next forward).  Call after editing a parameter IN PLACE
        (a run-time LoRA swap): fp8, merged or dequantised weights are copied into the fused buffers, not
        re-pointed, so such an edit is otherwise not seen.  `.to()`, `load_state_dict` and the fp8 switches do it
        themselves."""
        self._packed = None
        self.context_embedder._packed = None

    def invalidate_step_invariants(self):
        """Drop the cross-step caches (clean-latent embeddings, image projection, mask facts).  They are keyed by
        tensor identity and version counter; GPU tensors made under torch.inference_mode carry no version counter, so
        after writing INTO such a tensor between steps call this (`sampler.sample_hunyuan` calls it at the start of every
        section)."""
        self._invariant = {}

    def enable_fp8_compute(self, enable: bool = True):
        """Run the linears of the double/single blocks on the gfx950 fp8 MFMA (K6): weights held as e4m3 with a
        scale per output channel (a reference fp8 checkpoint's e4m3 weights and `scale_weight`, ref
        utils/fp8_optimization_utils.py, are used as they are), activations quantised to e4m3 per token row in
        front of every GEMM.  About twice the GEMM rate of bf16, but the activations lose bits the reference's
        W8A16 path keeps: a block's output moves by a few per cent relative L2, outside the 1e-2 parity bar, so
        this is off unless asked for.  Modulation, embedders, refiner and head stay bf16 (the reference
        quantises only `transformer_blocks` / `single_transformer_blocks`, excluding "norm")."""
        self.fp8_compute = bool(enable)
        self._packed = None
        self._ws = {}
        return self

    def enable_fp8_attention(self, enable: bool = True):
        """Opt-in: the joint attention of every block with Q, K, P and V in e4m3 (`fp_attn_fwd_fp8`; the role the
        reference gives its quantised SageAttention branch, hunyuan_video_packed.py:92-94).  A few per cent away
        from the bf16 result: outside the 1e-2 parity bar, never on by default."""
        self.fp8_attention = bool(enable)
        return self

    def _qk_attend(self, q, k, v, wq, wk, rope, n_img, wq_txt=None, wk_txt=None, normed=False, overlap=None):
        """K2 (RMSNorm of q and k per head, RoPE on the image rows) and the joint attention over [image rows | text
        rows]; output written over q.  Under sequence parallelism K2 writes its results and v straight into the send
        buffer of the head exchange (q, k, v themselves stay raw: only the attention output lands in q)."""
        attn = ops.attention_fp8 if self.fp8_attention else ops.attention
        if not self.sp_on:
            if not normed:                      # (normed: K2 was applied in the producing GEMM's epilogue, _ln_lin_qk)
                ops.qk_rmsnorm_rope_(q, k, wq, wk, rope, n_img, self.heads, wq_txt=wq_txt, wk_txt=wk_txt)
            attn(q, k, v, self.heads, out=q)
        else:
            from .ulysses import pick_chunks, ulysses_attention
            hl = self.heads // self.sp_world
            chunks = self.sp_chunks if self.sp_chunks is not None else pick_chunks(self._sp_total, hl)
            from .ulysses import exchange_buffers
            hwc = hl // chunks * self.head_dim
            send = exchange_buffers(self._sp_bufs, chunks, self.sp_world, q.shape[0], hwc, q.dtype, q.device)["send"]
            packed = ops.qk_rmsnorm_rope_pack(q, k, v, wq, wk, rope, n_img, self.heads, self.sp_world, chunks,
                                              wq_txt=wq_txt, wk_txt=wk_txt, out=send)
            ulysses_attention(q, k, v, self._sp_total, self.heads, self.head_dim,
                              lambda jq, jk, jv, h, o: attn(jq, jk, jv, h, out=o), self.sp_group, out=q,
                              chunks=chunks, packed=packed, overlap=overlap, bufs=self._sp_bufs)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, **_):
        """Local directory with config.json + *.safetensors (ref framepack/models.py:234-238 loads
        lllyasviel/FramePackI2V_HY this way; hub names are not resolvable offline)."""
        from safetensors.torch import load_file
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: from_pretrained needs a local model directory (no network)")
        with open(os.path.join(path, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**cfg)
        sd = {}
        for name in sorted(os.listdir(path)):
            if name.endswith(".safetensors"):
                sd.update(load_file(os.path.join(path, name)))
        model.load_state_dict(sd, strict=True)
        return model.to(torch_dtype) if torch_dtype is not None else model

    # ---- weight fusion ---------------------------------------------------------------------
    def _apply(self, fn, *a, **k):                      # .to() / .cuda() / .half(): fused buffers go stale
        self.invalidate_packed()
        self._invariant = {}
        self._ws = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        self._invariant = {}
        return super().load_state_dict(*a, **k)

    @staticmethod
    def _dense_weight(lin: nn.Linear) -> torch.Tensor:
        """bf16 weight of a Linear; FP8 layers (E4M3 weight + `scale_weight` buffer, ref
        utils/fp8_optimization_utils.py:216-226) are dequantised once here: numerically the reference's
        W8A16 path (dequantise, then a bf16 GEMM), without repeating the dequantisation every call."""
        w = lin.weight.detach()
        sw = getattr(lin, "scale_weight", None)
        if sw is not None:
            return (w.to(sw.dtype) * sw).to(BF16)
        return w.to(BF16)

    def _fuse_fp8(self, linears):
        """(e4m3 weight [N,K], fp32 scale [N]) and the bf16 bias of the row-concatenated linears."""
        qs, scs = [], []
        for l in linears:
            q, sc = ops.quantize_weight_fp8(l.weight.detach(), getattr(l, "scale_weight", None))
            qs.append(q)
            scs.append(sc)
        b = torch.cat([l.bias.detach().to(BF16) for l in linears], dim=0).contiguous()
        return (torch.cat(qs, dim=0).contiguous(), torch.cat(scs, dim=0).contiguous()), b

    def _fuse(self, linears, block: bool = False):
        if block and self.fp8_compute:
            return self._fuse_fp8(linears)
        ws = [self._dense_weight(l) for l in linears]
        W = torch.cat(ws, dim=0).contiguous()
        b = torch.cat([l.bias.detach().to(BF16) for l in linears], dim=0).contiguous()
        # re-point plain bf16 parameters at slices of the fused buffer: no second copy stays alive
        r0 = 0
        for l, w in zip(linears, ws):
            n = w.shape[0]
            if l.weight.dtype == BF16 and getattr(l, "scale_weight", None) is None:
                l.weight.data = W[r0:r0 + n]
                l.bias.data = b[r0:r0 + n]
            r0 += n
        return W, b

    def _pack(self):
        # callers run under torch.inference_mode() (ref k_diffusion_hunyuan.py:27): the fused buffers must be
        # ordinary tensors, because parameters are re-pointed at them
        with torch.inference_mode(False), torch.no_grad():
            return self._pack_impl()

    def _pack_impl(self):
        P = SimpleNamespace(double=[], single=[])
        mods = []
        for blk in self.transformer_blocks:
            a = blk.attn
            e = SimpleNamespace()
            e.w_qkv, e.b_qkv = self._fuse([a.to_q, a.to_k, a.to_v], block=True)
            e.w_qkv_c, e.b_qkv_c = self._fuse([a.add_q_proj, a.add_k_proj, a.add_v_proj], block=True)
            e.w_out, e.b_out = self._fuse([a.to_out[0]], block=True)
            e.w_out_c, e.b_out_c = self._fuse([a.to_add_out], block=True)
            e.w_ff1, e.b_ff1 = self._fuse([blk.ff.net[0].proj], block=True)
            e.w_ff2, e.b_ff2 = self._fuse([blk.ff.net[2]], block=True)
            e.w_ff1_c, e.b_ff1_c = self._fuse([blk.ff_context.net[0].proj], block=True)
            e.w_ff2_c, e.b_ff2_c = self._fuse([blk.ff_context.net[2]], block=True)
            e.nq, e.nk = a.norm_q.weight.detach().to(BF16).contiguous(), a.norm_k.weight.detach().to(BF16).contiguous()
            e.nq_c = a.norm_added_q.weight.detach().to(BF16).contiguous()
            e.nk_c = a.norm_added_k.weight.detach().to(BF16).contiguous()
            P.double.append(e)
            mods += [blk.norm1.linear, blk.norm1_context.linear]
        for blk in self.single_transformer_blocks:
            a = blk.attn
            e = SimpleNamespace()
            e.w_in, e.b_in = self._fuse([a.to_k, a.to_v, a.to_q, blk.proj_mlp], block=True)       # [k | v | q | mlp]
            e.w_out, e.b_out = self._fuse([blk.proj_out], block=True)
            e.nq, e.nk = a.norm_q.weight.detach().to(BF16).contiguous(), a.norm_k.weight.detach().to(BF16).contiguous()
            P.single.append(e)
            mods.append(blk.norm.linear)
        mods.append(self.norm_out.linear)
        # every AdaLN modulation of the forward depends only on temb: ONE weight-streaming GEMV per step
        P.w_mod, P.b_mod = self._fuse(mods)
        P.w_xe = self.x_embedder.proj.weight.detach().to(BF16).flatten(1).contiguous()
        P.b_xe = self.x_embedder.proj.bias.detach().to(BF16).contiguous()
        if self.clean_x_embedder is not None:
            P.w_ce = [getattr(self.clean_x_embedder, n).weight.detach().to(BF16).flatten(1).contiguous()
                      for n in ("proj", "proj_2x", "proj_4x")]
            P.b_ce = [getattr(self.clean_x_embedder, n).bias.detach().to(BF16).contiguous()
                      for n in ("proj", "proj_2x", "proj_4x")]
        self._packed = P
        return P

    def _workspace(self, S: int, device):
        """Views [S, ...] of a grow-only backing store per device: a negative prompt (another text length, so another
        S) alternates with the positive one every step without freeing and re-allocating ~1 GB twice per step."""
        key = str(device)
        back = self._ws.get(key)
        d = self.inner_dim
        m = self.transformer_blocks[0].ff.net[0].proj.out_features if len(self.transformer_blocks) else 4 * d
        if back is None or back.rows < S:
            rows = S if back is None else max(S, back.rows)
            self._ws.pop(key, None)
            back = None                                   # release before the larger allocation
            n_mod = (12 * len(self.transformer_blocks) + 3 * len(self.single_transformer_blocks) + 2) * d
            back = SimpleNamespace(rows=rows, X=torch.empty((rows, d), dtype=BF16, device=device),
                                   N=torch.empty((rows, d), dtype=BF16, device=device),
                                   Y=torch.empty((rows, 3 * d + m), dtype=BF16, device=device),
                                   mods=torch.empty((n_mod,), dtype=BF16, device=device),
                                   acc=ops.rel_l1_acc(device), views={})
            if self.fp8_compute:                              # e4m3 image of one GEMM's A operand + row scales
                back.Q = torch.empty((rows * (d + m),), dtype=ops.FP8, device=device)
                back.QS = torch.empty((rows,), dtype=torch.float32, device=device)
            self._ws[key] = back
        ws = back.views.get(S)
        if ws is None:
            ws = SimpleNamespace(X=back.X[:S], N=back.N[:S], Y=back.Y[:S], mods=back.mods, acc=back.acc)
            if self.fp8_compute:
                ws.Q, ws.QS = back.Q, back.QS
            back.views[S] = ws
        return ws

    # ---- packing of the input streams (ref :822-879) ----------------------------------------
    def _embed_stream(self, x, kernel, w, b, out):
        rows = patch_rows(pad_to_multiple(x, kernel), kernel)
        return ops.gemm(rows, w, b, out=out)

    def process_input_hidden_states(self, latents, latent_indices=None, clean_latents=None, clean_latent_indices=None,
                                    clean_latents_2x=None, clean_latent_2x_indices=None, clean_latents_4x=None,
                                    clean_latent_4x_indices=None, out: Optional[torch.Tensor] = None):
        """Returns (hidden [B,Si,d], rope [B,Si,256] fp32), token order [4x | 2x | 1x clean | noisy]."""
        P = self._packed or self._pack()
        B, C, T, H, W = latents.shape
        assert B == 1
        pt, p = self.config["patch_size_t"], self.config["patch_size"]
        t, h, w = T // pt, H // p, W // p
        if latent_indices is None:
            latent_indices = torch.arange(0, t).unsqueeze(0).expand(B, -1)
        streams = []     # (tensor, kernel, weight idx, indices, pool) in FINAL order
        if clean_latents_4x is not None and clean_latent_4x_indices is not None:
            streams.append((clean_latents_4x, (4, 8, 8), 2, clean_latent_4x_indices, 4))
        if clean_latents_2x is not None and clean_latent_2x_indices is not None:
            streams.append((clean_latents_2x, (2, 4, 4), 1, clean_latent_2x_indices, 2))
        if clean_latents is not None and clean_latent_indices is not None:
            streams.append((clean_latents, (1, 2, 2), 0, clean_latent_indices, 1))
        streams.append((latents, (pt, p, p), -1, latent_indices, 1))

        def ntok(x, k):
            return -(-x.shape[2] // k[0]) * -(-x.shape[3] // k[1]) * -(-x.shape[4] // k[2])

        counts = [ntok(x, k) for x, k, *_ in streams[:-1]] + [t * h * w]
        Si = sum(counts)
        if out is None:
            out = torch.empty((Si, self.inner_dim), dtype=BF16, device=latents.device)
        r0 = 0
        for (x, k, wi, _, _), n in zip(streams, counts):
            if wi < 0:                                                  # Conv3d floors a ragged edge (no padding)
                x = x.to(latents)[:, :, :t * pt, :h * p, :w * p]
                self._embed_stream(x, k, P.w_xe, P.b_xe, out[r0:r0 + n])
            else:                                                       # clean streams: the same tensors all section long
                emb = self._cached(f"clean{wi}", x, lambda x=x, k=k, wi=wi: self._embed_stream(
                    x.to(latents), k, P.w_ce[wi], P.b_ce[wi], None))    # ref :841, 852, 866
                out[r0:r0 + n].copy_(emb)
            r0 += n
        rope = self._rope_cache.get([(i, pool) for _, _, _, i, pool in streams], h, w, latents.device)
        return out[:Si].unsqueeze(0), rope

    # ---- the blocks -------------------------------------------------------------------------
    @staticmethod
    def _lin(ws, a, w, b, **kw):
        """One block linear: bf16 GEMM, or (fp8 compute) row quantisation + fp8 GEMM."""
        if not isinstance(w, tuple):
            return ops.gemm(a, w, b, **kw)
        M, K = a.shape
        aq, asc = ops.quantize_rows_fp8(a, out=ws.Q[:M * K].view(M, K), scale=ws.QS[:M])
        return ops.gemm_fp8(aq, asc, w[0], w[1], b, **kw)

    @staticmethod
    def _ln_lin(ws, x, sc, sh, n, w, b, **kw):
        """LayerNorm+modulate of the rows x followed by one block linear.  bf16: K3 into n, then the GEMM.  fp8 compute:
        the fused K3 + row quantiser writes the GEMM's e4m3 operand directly (n is not touched)."""
        if not isinstance(w, tuple):
            ops.layernorm_modulate(x, sc, sh, out=n)
            return ops.gemm(n, w, b, **kw)
        M, K = x.shape
        aq, asc = ops.layernorm_modulate_fp8(x, sc, sh, out=ws.Q[:M * K].view(M, K), out_scale=ws.QS[:M])
        return ops.gemm_fp8(aq, asc, w[0], w[1], b, **kw)

    def _fuse_qk(self, w, out) -> bool:
        """K2 in the q | k | v GEMM's epilogue (ops.gemm_qknorm): one GPU, bf16 weights, output aligned for the 4-wave
        kernel; off by default (`fuse_qk_norm`, FRAMEPACK_FUSE_QK=1: see __init__)."""
        return (self.fuse_qk_norm and not self.sp_on and not isinstance(w, tuple) and w.dtype == torch.bfloat16
                and ops.gemm_qknorm_ok(out))

    def _ln_lin_qk(self, ws, x, sc, sh, n, w, b, out, wq, wk, rope, rope_rows, qcol0, kcol0, **kw):
        """_ln_lin for a projection whose output holds q and k: with K2 fused into the GEMM when `_fuse_qk` allows.
        Returns whether q and k left the GEMM normalised."""
        if not self._fuse_qk(w, out):
            self._ln_lin(ws, x, sc, sh, n, w, b, out=out, **{k_: v_ for k_, v_ in kw.items() if k_ in ("epilogue", "gelu_from")})
            return False
        ops.layernorm_modulate(x, sc, sh, out=n)
        ops.gemm_qknorm(n, w, b, out, wq, wk, rope, rope_rows, qcol0, kcol0, self.heads, **kw)
        return True

    def _double_block(self, e, mod_i, mod_c, ws, Si, S, rope, side=None):
        """One double-stream block (ref :537-604).  The two streams meet only in the joint attention: everything else of
        the text stream (its rows of X, N, Y; its own weights) is independent of the image stream's.  With `side`
        (= (stream, event after the text stream's last kernel so far); `_run_blocks`) the text stream's kernels go to
        that low-priority HIP stream: its GEMMs have 48 - 192 tiles for 256 CUs and run in the CUs that the image
        stream's GEMMs leave idle in their last, partly filled round instead of in launches of their own.  Two joins per
        block: the attention waits for the text q | k | v, the text out-projection for the attention."""
        d, H = self.inner_dim, self.heads
        X, N, Y = ws.X, ws.N, ws.Y
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod_i.chunk(6)
        csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = mod_c.chunk(6)
        qkv = Y[:, :3 * d]
        q, k, v = Y[:, :d], Y[:, d:2 * d], Y[:, 2 * d:3 * d]
        Fh = Y[:, 3 * d:]
        GR = ops.EPI_GATE_RESIDUAL

        normed = [False, False]

        def text_pre():                 # (text rows: their own norm weights, no RoPE)
            normed[1] = self._ln_lin_qk(ws, X[Si:], csc_a, csh_a, N[Si:], e.w_qkv_c, e.b_qkv_c, qkv[Si:], e.nq_c, e.nk_c,
                                        None, 0, 0, d)

        def text_post():
            self._lin(ws, q[Si:], e.w_out_c, e.b_out_c, out=X[Si:], epilogue=GR, resid=X[Si:], gate=cg_a)
            self._ln_lin(ws, X[Si:], csc_m, csh_m, N[Si:], e.w_ff1_c, e.b_ff1_c, out=Fh[Si:], epilogue=ops.EPI_GELU)
            self._lin(ws, Fh[Si:], e.w_ff2_c, e.b_ff2_c, out=X[Si:], epilogue=GR, resid=X[Si:], gate=cg_m)

        if self._group_rows(e, Si, S):
            # both streams' linears as one launch per pair: the text segment's 4 - 16 row tiles ride in the image
            # segment's partly filled last round (bit-identical to the separate launches)
            ops.layernorm_modulate(X[:Si], sc_a, sh_a, out=N[:Si])
            ops.layernorm_modulate(X[Si:], csc_a, csh_a, out=N[Si:])
            ops.gemm_rows2(N, Si, e.w_qkv, e.w_qkv_c, e.b_qkv, e.b_qkv_c, out=qkv)
            self._qk_attend(q, k, v, e.nq, e.nk, rope, Si, wq_txt=e.nq_c, wk_txt=e.nk_c, normed=False)
            ops.gemm_rows2(q, Si, e.w_out, e.w_out_c, e.b_out, e.b_out_c, out=X, epilogue=GR, resid=X, gate0=g_a, gate1=cg_a)
            ops.layernorm_modulate(X[:Si], sc_m, sh_m, out=N[:Si])
            ops.layernorm_modulate(X[Si:], csc_m, csh_m, out=N[Si:])
            ops.gemm_rows2(N, Si, e.w_ff1, e.w_ff1_c, e.b_ff1, e.b_ff1_c, out=Fh, epilogue=ops.EPI_GELU)
            ops.gemm_rows2(Fh, Si, e.w_ff2, e.w_ff2_c, e.b_ff2, e.b_ff2_c, out=X, epilogue=GR, resid=X, gate0=g_m, gate1=cg_m)
            return
        main = torch.cuda.current_stream(X.device) if side is not None else None
        normed[0] = self._ln_lin_qk(ws, X[:Si], sc_a, sh_a, N[:Si], e.w_qkv, e.b_qkv, qkv[:Si], e.nq, e.nk, rope, Si, 0, d)
        if side is None:
            text_pre()
        else:
            with torch.cuda.stream(side[0]):
                text_pre()
                side[1].record(side[0])
            main.wait_event(side[1])
        if normed[0] != normed[1]:          # (one stream fused, the other not: finish the other's K2 by itself)
            lo, hi = (Si, S) if normed[0] else (0, Si)
            ops.qk_rmsnorm_rope_(q[lo:hi], k[lo:hi], e.nq_c if normed[0] else e.nq, e.nk_c if normed[0] else e.nk,
                                 None if normed[0] else rope, 0 if normed[0] else Si, H)
        self._qk_attend(q, k, v, e.nq, e.nk, rope, Si, wq_txt=e.nq_c, wk_txt=e.nk_c, normed=normed[0] or normed[1])
        if side is not None:
            side[2].record(main)
            side[0].wait_event(side[2])
            with torch.cuda.stream(side[0]):
                text_post()
                side[1].record(side[0])
        self._lin(ws, q[:Si], e.w_out, e.b_out, out=X[:Si], epilogue=GR, resid=X[:Si], gate=g_a)
        if side is None:
            self._lin(ws, q[Si:], e.w_out_c, e.b_out_c, out=X[Si:], epilogue=GR, resid=X[Si:], gate=cg_a)
        self._ln_lin(ws, X[:Si], sc_m, sh_m, N[:Si], e.w_ff1, e.b_ff1, out=Fh[:Si], epilogue=ops.EPI_GELU)
        if side is None:
            self._ln_lin(ws, X[Si:], csc_m, csh_m, N[Si:], e.w_ff1_c, e.b_ff1_c, out=Fh[Si:], epilogue=ops.EPI_GELU)
        self._lin(ws, Fh[:Si], e.w_ff2, e.b_ff2, out=X[:Si], epilogue=GR, resid=X[:Si], gate=g_m)
        if side is None:
            self._lin(ws, Fh[Si:], e.w_ff2_c, e.b_ff2_c, out=X[Si:], epilogue=GR, resid=X[Si:], gate=cg_m)

    def _group_rows(self, e, Si, S) -> bool:
        """Whether a double block's linear pairs run as one launch each (ops.gemm_rows2): one GPU, bf16 weights (the fp8
        GEMM has no row-segment form), both segments non-empty, K2 as its own pass."""
        return (self.group_text_gemms and not self.sp_on and not self.fp8_compute and 0 < Si < S
                and not isinstance(e.w_qkv, tuple) and e.w_qkv.dtype == torch.bfloat16 and not self.fuse_qk_norm)

    def _single_block(self, e, mod, ws, Si, S, rope):
        d, H = self.inner_dim, self.heads
        X, N, Y = ws.X, ws.N, ws.Y
        shift, scale, gate = mod.chunk(3)
        if self.sp_on and self.sp_overlap_mlp and not isinstance(e.w_in, tuple):
            # sequence parallel: the in-projection as two GEMMs over the same normalised rows - [k | v | q] first, the
            # MLP half (57 % of the columns) while q | k | v are on their way to the owners of their heads.  Same
            # arithmetic per output element as the one launch.
            ops.layernorm_modulate(X, scale, shift, out=N)
            ops.gemm(N, e.w_in[:3 * d], e.b_in[:3 * d], out=Y[:, :3 * d])
            self._qk_attend(Y[:, 2 * d:3 * d], Y[:, :d], Y[:, d:2 * d], e.nq, e.nk, rope, Si,
                            overlap=lambda: ops.gemm(N, e.w_in[3 * d:], e.b_in[3 * d:], out=Y[:, 3 * d:],
                                                     epilogue=ops.EPI_GELU))
            self._lin(ws, Y[:, 2 * d:], e.w_out, e.b_out, out=X, epilogue=ops.EPI_GATE_RESIDUAL, resid=X, gate=gate)
            return
        normed = self._ln_lin_qk(ws, X, scale, shift, N, e.w_in, e.b_in, Y, e.nq, e.nk, rope, Si, 2 * d, 0,
                                 epilogue=ops.EPI_GELU_FROM, gelu_from=3 * d)               # [k | v | q | gelu(mlp)]
        k, v, q = Y[:, :d], Y[:, d:2 * d], Y[:, 2 * d:3 * d]
        self._qk_attend(q, k, v, e.nq, e.nk, rope, Si, normed=normed)
        self._lin(ws, Y[:, 2 * d:], e.w_out, e.b_out, out=X, epilogue=ops.EPI_GATE_RESIDUAL, resid=X, gate=gate)

    def _text_side_stream(self, device):
        """(low-priority stream, two events) for the text stream of the double blocks, per device; None where the
        overlap is off: fp8 compute (one shared quantiser scratch), sequence parallelism (rows are not split by stream
        there), block taps (they read X between blocks), `text_stream_overlap = False`."""
        if not self.text_stream_overlap or self.sp_on or self.taps is not None or self.fp8_compute:
            return None
        key = device.index
        if key not in self._side_streams:
            least = torch.cuda.Stream.priority_range()[0] if hasattr(torch.cuda.Stream, "priority_range") else 0
            self._side_streams[key] = (torch.cuda.Stream(device=device, priority=least), torch.cuda.Event(),
                                       torch.cuda.Event())
        return self._side_streams[key]

    def _run_blocks(self, P, mods, ws, Si, S, rope):
        d = self.inner_dim
        o = 0
        side = self._text_side_stream(ws.X.device) if len(P.double) and S > Si else None
        if side is not None and self._group_rows(P.double[0], Si, S):
            side = None
        if side is not None:                        # the side stream starts behind everything queued so far
            main = torch.cuda.current_stream(ws.X.device)
            side[2].record(main)
            side[0].wait_event(side[2])
        for i, e in enumerate(P.double):
            self._double_block(e, mods[o:o + 6 * d], mods[o + 6 * d:o + 12 * d], ws, Si, S, rope, side)
            o += 12 * d
            if self.taps is not None:
                self.taps[f"double.{i}"] = (ws.X[:Si].clone(), ws.X[Si:].clone())
        if side is not None:                        # the single blocks read all rows
            main.wait_event(side[1])
        for i, e in enumerate(P.single):
            self._single_block(e, mods[o:o + 3 * d], ws, Si, S, rope)
            o += 3 * d
            if self.taps is not None:
                self.taps[f"single.{i}"] = (ws.X[:Si].clone(), ws.X[Si:].clone())

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                guidance, latent_indices=None, clean_latents=None, clean_latent_indices=None, clean_latents_2x=None,
                clean_latent_2x_indices=None, clean_latents_4x=None, clean_latent_4x_indices=None,
                image_embeddings=None, attention_kwargs=None, return_dict=True):
        """ref :881-1062.  Same arguments, same result tuple / `.sample`, same exceptions."""
        shape = hidden_states.shape
        if len(shape) == 4:
            batch_size, num_frames, height, width = shape
        elif len(shape) == 5:
            batch_size, channels, num_frames, height, width = shape
        else:
            raise ValueError(f"Unexpected hidden_states shape: {shape}. Expected 4D (B,F,H,W) or 5D (B,C,F,H,W) tensor.")
        if not hidden_states.is_cuda:
            raise RuntimeError("HunyuanVideoTransformer3DModelPacked (MI355X build) runs on the GPU only; "
                               "there is no CPU fallback")
        if hidden_states.dtype != BF16 or self.dtype != BF16:
            raise TypeError("the gfx950 kernels compute in bf16: load the model and pass hidden_states as torch.bfloat16")
        _lib.load()
        if batch_size > 1:
            # ref :932-941 packs a batch with cu_seqlens (segments [image + valid text | padded text] per sample) and
            # needs a varlen kernel it may not have ("No Attn Installed!").  The rows it returns - the image tokens -
            # attend to their own sample's image and valid text tokens only, so running the samples one after the
            # other through the single-sample path returns exactly the same values, at full kernel efficiency
            # (every sample already fills the chip).  TeaCache keeps ONE state for the batch in the reference.
            if self.enable_teacache:
                raise NotImplementedError("TeaCache with batch size > 1 is not supported: run the samples separately")

            def cut(x, b):
                return None if x is None else x[b:b + 1]
            with torch.no_grad(), torch.cuda.device(hidden_states.device):
                outs = [self._forward(*(cut(x, b) for x in (
                    hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                    guidance, latent_indices, clean_latents, clean_latent_indices, clean_latents_2x,
                    clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices, image_embeddings)), False)[0]
                    for b in range(batch_size)]
            out = torch.cat(outs, dim=0)
            return Transformer2DModelOutput(sample=out) if return_dict else (out,)
        # (the kernels launch on the CURRENT device: a model on cuda:1 works without the caller's set_device)
        with torch.no_grad(), torch.cuda.device(hidden_states.device):
            return self._forward(hidden_states, timestep, encoder_hidden_states, encoder_attention_mask,
                                 pooled_projections, guidance, latent_indices, clean_latents, clean_latent_indices,
                                 clean_latents_2x, clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices,
                                 image_embeddings, return_dict)

    def _forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                 guidance, latent_indices, clean_latents, clean_latent_indices, clean_latents_2x,
                 clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices, image_embeddings, return_dict):
        dev = hidden_states.device
        B, C, T, Hh, Ww = hidden_states.shape
        p, p_t = self.config["patch_size"], self.config["patch_size_t"]
        t2, h2, w2 = T // p_t, Hh // p, Ww // p
        ctx_len = t2 * h2 * w2
        d = self.inner_dim
        P = self._packed or self._pack()

        # conditioning (tiny, once per step): temb, refined text context, image projection
        temb = self.time_text_embed.timestep_embedder(_sinusoid_256(timestep).to(pooled_projections.dtype)) \
            + self.time_text_embed.guidance_embedder(_sinusoid_256(guidance).to(pooled_projections.dtype)) \
            + self.time_text_embed.text_embedder(pooled_projections)
        # facts about the mask (valid count, prefix or not): read back once per mask tensor, not once per step
        if encoder_attention_mask is None:
            minfo = _MaskInfo(None)
        else:
            # (one entry per mask tensor: with a negative prompt two masks alternate every step)
            slot = f"mask_info@{encoder_attention_mask.data_ptr():x}"
            if slot not in self._invariant:
                for old in [k_ for k_ in self._invariant if k_.startswith("mask_info@")][3:]:
                    del self._invariant[old]
            minfo = self._cached(slot, encoder_attention_mask, lambda: _MaskInfo(encoder_attention_mask.to(dev)))
            encoder_attention_mask = encoder_attention_mask.to(dev)
        ctx = self.context_embedder(encoder_hidden_states, timestep, encoder_attention_mask, minfo)
        text_len = ctx.shape[1] if minfo.n_valid is None else minfo.n_valid
        if self.image_projection is not None:
            assert image_embeddings is not None, "You must use image embeddings!"
            extra = self._cached("image_projection", image_embeddings,
                                 lambda: self.image_projection(image_embeddings))
            ctx = torch.cat([extra, ctx], dim=1)                     # image tokens before text (ref :922-923)
            text_len += extra.shape[1]                               # its mask is all ones (ref :924)

        assert B == 1          # forward() feeds a batch one sample at a time
        ctx = ctx[:, :text_len]                                      # ref :930-931, without the per-step host sync

        # packed image tokens straight into the residual-stream buffer
        n_img = self._count_image_tokens(hidden_states, clean_latents, clean_latent_indices, clean_latents_2x,
                                         clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices)
        if not self.sp_on:
            lo, Si = 0, n_img
            S = Si + text_len
            ws = self._workspace(S, dev)
            _, rope = self.process_input_hidden_states(hidden_states, latent_indices, clean_latents,
                                                       clean_latent_indices, clean_latents_2x, clean_latent_2x_indices,
                                                       clean_latents_4x, clean_latent_4x_indices, out=ws.X[:n_img])
            rope = rope[0]
        else:
            # sequence parallel: this rank keeps rows [g0, g0 + valid) of the packed sequence [image | text]; the
            # rest of its n rows is padding (computed, never attended to).  Si = its image rows, the others are text.
            from .ulysses import row_shard
            S_total = n_img + text_len
            n_loc, g0, valid = row_shard(S_total, self.sp_world, self.sp_rank)
            lo = min(g0, n_img)
            Si = min(g0 + valid, n_img) - lo
            t0 = max(g0, n_img) - n_img
            tvalid = valid - Si
            self._sp_total = S_total
            S = n_loc
            ws = self._workspace(S, dev)
            full, rope = self.process_input_hidden_states(hidden_states, latent_indices, clean_latents,
                                                          clean_latent_indices, clean_latents_2x,
                                                          clean_latent_2x_indices, clean_latents_4x,
                                                          clean_latent_4x_indices)
            ws.X[:Si].copy_(full[0, lo:lo + Si])
            rope = rope[0, lo:lo + Si].contiguous()
            ctx = torch.cat([ctx[:, t0:t0 + tvalid], ctx.new_zeros((1, n_loc - valid, ctx.shape[2]))], dim=1)
        ws.X[Si:].copy_(ctx[0])

        # all AdaLN modulations of this step: one GEMV over the concatenated modulation weights
        mods = ops.silu_gemv(temb[0], P.w_mod, P.b_mod, out=ws.mods)

        if self.taps is not None:
            self.taps["packed_hidden"], self.taps["rope"] = ws.X[:Si].clone(), rope
            self.taps["temb"], self.taps["text_ctx"] = temb, ctx

        if self.enable_teacache:
            sh_a, sc_a = mods[:d], mods[d:2 * d]
            modulated = ops.layernorm_modulate(ws.X[:Si], sc_a, sh_a)                       # ref :945
            self.cache_queries += 1
            curr = 0.0
            if self.cnt == 0 or self.cnt == self.num_steps - 1:
                should_calc = True
                self.accumulated_rel_l1_distance = 0
                self.cache_misses += 1
            else:
                acc_t = ops.rel_l1(modulated, self.previous_modulated_input, ws.acc) if Si > 0 else ws.acc[:2].zero_()
                if self.sp_on:                              # identical decision on every rank
                    import torch.distributed as dist
                    dist.all_reduce(acc_t, group=self.sp_group)
                acc = acc_t.tolist()                                           # host sync (ref :958)
                n = n_img * d                               # elements over all ranks
                # the reference takes both means and their quotient in bf16 (0-dim bf16 tensors, ref :957-958)
                curr = _bf16_round(_bf16_round(acc[0] / n) / _bf16_round(_bf16_round(acc[1] / n) + 1e-8))
                self.accumulated_rel_l1_distance += self.teacache_rescale_func(curr)
                should_calc = self.accumulated_rel_l1_distance >= self.rel_l1_thresh
                if should_calc:
                    self.accumulated_rel_l1_distance = 0
                    self.cache_misses += 1
                else:
                    self.cache_hits += 1
            self.previous_modulated_input = modulated
            self.cnt += 1
            if self.cnt == self.num_steps:
                self.cnt = 0
            self.cache_info = {"hits": self.cache_hits, "misses": self.cache_misses, "queries": self.cache_queries,
                               "hit_rate": self.cache_hits / max(1, self.cache_queries), "step": self.cnt,
                               "curr_rel_l1": curr, "accumulated_distance": self.accumulated_rel_l1_distance}
            if not should_calc:
                ws.X[:Si].add_(self.previous_residual)
            else:
                ori = ws.X[:Si].clone()
                self._run_blocks(P, mods, ws, Si, S, rope)
                self.previous_residual = ws.X[:Si] - ori
        else:
            self._run_blocks(P, mods, ws, Si, S, rope)

        # output head (ref :1044-1057): norm_out is token-wise, so slice first; (scale, shift) order of :501
        scale, shift = mods[-2 * d:-d], mods[-d:]
        if not self.sp_on:
            # the noisy-latent tokens are the last ctx_len rows of the image sequence
            first = n_img - ctx_len
            head_in = ops.layernorm_modulate(ws.X[first:Si], scale, shift, out=ws.N[:Si - first])
        else:
            # every rank runs the (tiny) head over all of its rows; the noisy-latent rows are cut out of the gather
            head_in = ops.layernorm_modulate(ws.X, scale, shift, out=ws.N)
        if self.high_quality_fp32_output_for_inference:
            # ref :1048-1053: activations and proj_out's weights upcast to fp32 (in place, at the first call), then the
            # Linear.  The operands are then bf16 VALUES in fp32 containers, whose products are exact in fp32: K5 with the
            # accumulators written out in fp32 computes the same sum.  Only a proj_out that was loaded with genuine fp32
            # weights (never the product's bf16 checkpoint) cannot go that way and takes torch's fp32 Linear.
            if self.proj_out.weight.dtype != torch.float32:
                self.proj_out.to(dtype=torch.float32)
                self._packed = P          # proj_out is not part of the fused buffers
            head = getattr(P, "head32", None)       # (fp32 weights it was made from - held, so their address stays
            w32 = self.proj_out.weight.detach()     #  theirs -, bf16 copy, whether the copy is exact)
            ver = None if w32.is_inference() else w32._version      # (made under inference_mode: no version counter)
            if head is None or head[0].data_ptr() != w32.data_ptr() or head[3] != ver:
                w16 = w32.to(BF16).contiguous()
                head = P.head32 = (w32, w16, bool((w16.float() == w32).all()), ver)
            if head[2]:
                out = ops.gemm_f32out(head_in, head[1], self.proj_out.bias.detach().float().contiguous())
            else:
                out = F.linear(head_in.float(), self.proj_out.weight, self.proj_out.bias)
        else:
            out = ops.gemm(head_in, self.proj_out.weight.detach(), self.proj_out.bias.detach())
        if self.sp_on:
            from .ulysses import gather_rows
            out = gather_rows(out, self.sp_group)[n_img - ctx_len:n_img]
        if self.taps is not None:
            self.taps["head"] = out
        out = unpatchify(out, B, t2, h2, w2, self.config["out_channels"], p_t, p)
        if return_dict:
            return Transformer2DModelOutput(sample=out)
        return (out,)

    def _count_image_tokens(self, latents, c1, i1, c2, i2, c4, i4) -> int:
        p, p_t = self.config["patch_size"], self.config["patch_size_t"]
        _, _, T, H, W = latents.shape
        n = (T // p_t) * (H // p) * (W // p)

        def cnt(x, k):
            return -(-x.shape[2] // k[0]) * -(-x.shape[3] // k[1]) * -(-x.shape[4] // k[2])
        if c1 is not None and i1 is not None:
            n += cnt(c1, (1, 2, 2))
        if c2 is not None and i2 is not None:
            n += cnt(c2, (2, 4, 4))
        if c4 is not None and i4 is not None:
            n += cnt(c4, (4, 8, 8))
        return n
