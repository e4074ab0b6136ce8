"""Host mirror of the VAE boundary: `AutoencoderKLHunyuanVideo.decode` + `vae_decode`, `.encode` + `vae_encode`.

The reference's code for this path is ref: diffusers_helper/hunyuan.py:93-103 (`vae_decode`) and :61-89
(`vae_decode_fake`); the model class itself is diffusers 0.33.1's `AutoencoderKLHunyuanVideo`
(ref: framepack/models.py:158-163 constructs it with `from_pretrained(..., subfolder='vae', torch_dtype=float16)`).
This module keeps that class name, its parameter names (`decoder.*`, `post_quant_conv.*`, so a real checkpoint's
decoder half loads unchanged), `.config.scaling_factor`, `.dtype/.device` and `decode(z).sample`, and runs the
decoder on the gfx950 kernels of csrc/vae_kernels.hip: volumes are fp16 channels-last [T, H, W, C], every 3x3x3
convolution is an implicit GEMM, nearest upsampling is folded into the following convolution's gather,
GroupNorm+SiLU is one fused pass after a deterministic statistics pass, the resnet skip is added in the
convolution's epilogue, and the mid-block attention (one head of 512 channels, frame-causal) runs per frame and
chunk of query rows as fp32-logit GEMM (against the keys that frame may see) -> softmax into fp16 in place -> P.V GEMM
(ops.attention_framecausal): fp32 logits, and at most 512 MB of them at a time whatever the window's size.

The encode half (ref: diffusers_helper/hunyuan.py:107-110 `vae_encode`, run on single images: worker.py:246-272)
uses the same kernels: the downsamplers are the same implicit GEMM with a stride of 2 in its gather
(`fp_conv3d_causal_strided_f16`); parameter names `encoder.*` / `quant_conv.*` as in a diffusers checkpoint.

PARITY UNPINNED: the diffusers source is not available offline; the algorithm is the one restated in
oracle/vae_oracle.py (see its header).  There is no CPU path: `decode` / `encode` raise for CPU tensors.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

F16 = torch.float16


class _CausalConv3d(nn.Module):
    """Parameter container with diffusers' names: `<name>.conv.{weight,bias}`."""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, k)


class _Resnet(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = _CausalConv3d(cin, cout, 3)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = _CausalConv3d(cout, cout, 3)
        if cin != cout:
            self.conv_shortcut = _CausalConv3d(cin, cout, 1)


class _Attention(nn.Module):
    def __init__(self, c: int, groups: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])


class _MidBlock(nn.Module):
    def __init__(self, c: int, groups: int, add_attention: bool):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])
        self.attentions = nn.ModuleList([_Attention(c, groups) if add_attention else None])


class _Upsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = _CausalConv3d(c, c, 3)


class _UpBlock(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, groups: int, factor: Optional[Tuple[int, int, int]]):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        self.factor = factor
        if factor is not None:
            self.upsamplers = nn.ModuleList([_Upsample(cout)])


class _Downsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = _CausalConv3d(c, c, 3)


class _DownBlock(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, groups: int, stride: Optional[Tuple[int, int, int]]):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        self.stride = stride
        if stride is not None:
            self.downsamplers = nn.ModuleList([_Downsample(cout)])


class _Encoder(nn.Module):
    def __init__(self, in_ch: int, latent: int, chans: Tuple[int, ...], layers: int, groups: int, add_attention: bool,
                 plan: List[Tuple[int, int, Optional[Tuple[int, int, int]]]]):
        super().__init__()
        self.conv_in = _CausalConv3d(in_ch, chans[0], 3)
        self.down_blocks = nn.ModuleList([_DownBlock(ci, co, layers, groups, s) for ci, co, s in plan])
        self.mid_block = _MidBlock(chans[-1], groups, add_attention)
        self.conv_norm_out = nn.GroupNorm(groups, chans[-1], eps=1e-6)
        self.conv_out = _CausalConv3d(chans[-1], 2 * latent, 3)


class _Decoder(nn.Module):
    def __init__(self, latent: int, out_ch: int, chans: Tuple[int, ...], layers: int, groups: int, add_attention: bool,
                 plan: List[Tuple[int, int, Optional[Tuple[int, int, int]]]]):
        super().__init__()
        self.conv_in = _CausalConv3d(latent, chans[-1], 3)
        self.mid_block = _MidBlock(chans[-1], groups, add_attention)
        self.up_blocks = nn.ModuleList([_UpBlock(ci, co, layers + 1, groups, f) for ci, co, f in plan])
        self.conv_norm_out = nn.GroupNorm(groups, chans[0], eps=1e-6)
        self.conv_out = _CausalConv3d(chans[0], out_ch, 3)


def _up_block_plan(chans: Tuple[int, ...], spatial_ratio: int, temporal_ratio: int):
    """Which up block upsamples what (diffusers HunyuanVideoDecoder3D.__init__): with ratios 8/4 and four blocks,
    block 0 doubles H,W, blocks 1 and 2 double T,H,W, block 3 nothing."""
    import math
    rev = list(reversed(chans))
    n, n_sp, n_t = len(rev), int(math.log2(spatial_ratio)), int(math.log2(temporal_ratio))
    plan, out_ch = [], rev[0]
    for i in range(n):
        prev, out_ch = out_ch, rev[i]
        sp = i < n_sp
        tm = (i >= n - 1 - n_t) and i != n - 1
        plan.append((prev, out_ch, ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None))
    return plan


def _down_block_plan(chans: Tuple[int, ...], spatial_ratio: int, temporal_ratio: int):
    """Which down block strides what (diffusers HunyuanVideoEncoder3D.__init__): with ratios 8/4 and four blocks,
    block 0 halves H,W, blocks 1 and 2 halve T,H,W, block 3 nothing."""
    import math
    n, n_sp, n_t = len(chans), int(math.log2(spatial_ratio)), int(math.log2(temporal_ratio))
    plan, out_ch = [], chans[0]
    for i in range(n):
        prev, out_ch = out_ch, chans[i]
        sp = i < n_sp
        tm = (i >= n - 1 - n_t) and i != n - 1
        plan.append((prev, out_ch, ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None))
    return plan


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# FRAMEPACK_VAE_HALO_OVERLAP=0: the tiled decode exchanges the border rows BEFORE each convolution and computes on the
# extended band (round 2's form) instead of overlapping the exchange with the convolution (_conv_b)
_HALO_OVERLAP = os.environ.get("FRAMEPACK_VAE_HALO_OVERLAP", "1") != "0"


class DecoderOutput(SimpleNamespace):
    pass


class _Bands:
    """Spatial tiling of one decode over the ranks of a group: rank r owns a band of consecutive rows of H (at every
    resolution of the decoder: a band of n latent rows is 8 n image rows).  What crosses band borders:
      * a 3x3x3 convolution reads one row above and one below its output row (of its own input resolution): before
        every convolution the border rows are exchanged with the two neighbours (`extend`), the convolution runs on
        the extended band and the rows computed from the halo alone are cut off.  At the top and bottom of the volume
        there is no neighbour and the kernel's clamp is the replicate padding, as on one GPU;
      * GroupNorm statistics are global over (T, H, W): every rank reduces its own rows, the per-group (sum, sum of
        squares, count) are all-reduced (3 x groups doubles) and every rank normalises with the same (mean, rstd);
      * the mid-block attention sees all tokens: the k | v bands are all-gathered into (t, h, w) order, each rank
        keeps the queries of its band.
    Time is not split (the causal convolutions need every earlier frame) and neither is W."""

    def __init__(self, group, H: int, cache: Optional[dict] = None):
        import torch.distributed as dist
        self.group = group
        self.cache = cache                 # halo buffers, allocated once per shape (comm.neighbour_exchange_start)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if H < self.world:
            raise ValueError(f"{H} latent rows cannot be tiled over {self.world} ranks")
        base, rem = divmod(H, self.world)
        self.sizes = [base + (r < rem) for r in range(self.world)]            # latent rows per rank
        self.lo = sum(self.sizes[:self.rank])
        self.hi = self.lo + self.sizes[self.rank]
        self.top, self.bot = self.rank > 0, self.rank < self.world - 1

    def extend(self, x: torch.Tensor) -> torch.Tensor:
        """[T, Hb, W, C] -> [T, (1) + Hb + (1), W, C]: the neighbours' border rows around the band."""
        from .comm import neighbour_exchange
        fu, fd = neighbour_exchange(x[:, :1] if self.top else None, x[:, -1:] if self.bot else None, self.group)
        parts = ([fu] if self.top else []) + [x] + ([fd] if self.bot else [])
        return torch.cat(parts, dim=1) if len(parts) > 1 else x

    def exchange_start(self, x: torch.Tensor):
        """Post the exchange of the band's border rows; `.wait()` -> (row from above, row from below)."""
        from .comm import neighbour_exchange_start
        return neighbour_exchange_start(x[:, :1] if self.top else None, x[:, -1:] if self.bot else None, self.group,
                                        self.cache)

    def crop(self, y_ext: torch.Tensor, up_h: int) -> torch.Tensor:
        """Cut the rows computed from a halo row off a convolution's output (up_h output rows per input row)."""
        r0 = up_h if self.top else 0
        r1 = y_ext.shape[1] - (up_h if self.bot else 0)
        return y_ext[:, r0:r1].contiguous() if (self.top or self.bot) else y_ext

    def global_stats(self, stats: torch.Tensor, n_local: int, eps: float) -> torch.Tensor:
        """Per-group (mean, rstd) of this rank's rows -> (mean, rstd) over all ranks' rows.  The moments are rebuilt
        from (mean, rstd) in fp64: sum = mean n, sum of squares = (var + mean^2) n."""
        from .comm import all_reduce_sum
        st = stats.view(-1, 2).double()
        mean, rstd = st[:, 0], st[:, 1]
        var = (1.0 / (rstd * rstd) - eps).clamp_min(0.0)
        pack = torch.stack([mean * n_local, (var + mean * mean) * n_local, torch.full_like(mean, float(n_local))])
        all_reduce_sum(pack, self.group)
        n = pack[2]
        m = pack[0] / n
        v = (pack[1] / n - m * m).clamp_min(0.0)
        return torch.stack([m, 1.0 / torch.sqrt(v + eps)], dim=1).float().reshape(-1).contiguous()

    def gather_rows(self, x: torch.Tensor, rows_per_latent_row: int) -> torch.Tensor:
        """[T, Hb, W, C] bands of all ranks -> the whole volume [T, H, W, C] on every rank."""
        from .comm import all_gather_padded
        parts = all_gather_padded(x.permute(1, 0, 2, 3).contiguous(), [n * rows_per_latent_row for n in self.sizes],
                                  self.group)
        return torch.cat(parts, dim=0).permute(1, 0, 2, 3)


class DiagonalGaussianDistribution:
    """What `vae.encode(x).latent_dist` offers in diffusers: parameters [B, 2C, ...] = (mean | logvar), the log
    variance clamped to [-30, 20]; `sample(generator)` = mean + std * N(0, 1), `mode()` = mean."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput(SimpleNamespace):
    pass


class AutoencoderKLHunyuanVideo(nn.Module):
    """diffusers' AutoencoderKLHunyuanVideo on MI355X kernels (see module docstring)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 16,
                 block_out_channels: Tuple[int, ...] = (128, 256, 512, 512), layers_per_block: int = 2,
                 norm_num_groups: int = 32, scaling_factor: float = 0.476986, spatial_compression_ratio: int = 8,
                 temporal_compression_ratio: int = 4, mid_block_add_attention: bool = True):
        super().__init__()
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            norm_num_groups=norm_num_groups, scaling_factor=scaling_factor,
            spatial_compression_ratio=spatial_compression_ratio, temporal_compression_ratio=temporal_compression_ratio,
            mid_block_add_attention=mid_block_add_attention)
        plan = _up_block_plan(tuple(block_out_channels), spatial_compression_ratio, temporal_compression_ratio)
        self.decoder = _Decoder(latent_channels, out_channels, tuple(block_out_channels), layers_per_block,
                                norm_num_groups, mid_block_add_attention, plan)
        self.encoder = _Encoder(in_channels, latent_channels, tuple(block_out_channels), layers_per_block,
                                norm_num_groups, mid_block_add_attention,
                                _down_block_plan(tuple(block_out_channels), spatial_compression_ratio,
                                                 temporal_compression_ratio))
        self.quant_conv = nn.Conv3d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv3d(latent_channels, latent_channels, 1)
        self._encoder_loaded = True        # random-init counts as loaded; a decoder-only checkpoint clears it
        # diffusers defaults: framewise (temporally tiled) decoding on, spatial tiling / slicing off
        self.use_framewise_decoding = True
        self.tile_sample_min_num_frames = 16
        self.tile_sample_stride_num_frames = 12
        self._prep: Optional[Dict[str, torch.Tensor]] = None
        self._prep_key = None

    # ---- reference-visible attributes -----------------------------------------------------------------------
    @property
    def dtype(self) -> torch.dtype:
        return self.post_quant_conv.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.post_quant_conv.weight.device

    def enable_slicing(self) -> None:      # low-VRAM knobs of the reference (models.py:280-282): not needed here
        pass

    def enable_tiling(self) -> None:
        pass

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """A full diffusers checkpoint holds both halves.  A decoder-only state dict (no `encoder.*` / `quant_conv.*`
        key at all) loads strictly into the decode half and leaves `encode` unavailable."""
        self._prep = None
        enc_keys = [k for k in state_dict if k.startswith("encoder.") or k.startswith("quant_conv.")]
        if enc_keys or not strict:
            self._encoder_loaded = bool(enc_keys)
            return super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._encoder_loaded = False
        mine = {k: v for k, v in self.state_dict().items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
        return super().load_state_dict({**mine, **state_dict}, strict=True, assign=assign)

    def _apply(self, fn, *a, **kw):
        self._prep = None
        return super()._apply(fn, *a, **kw)

    # ---- weight preparation -----------------------------------------------------------------------------------
    def _prepare(self) -> Dict[str, torch.Tensor]:
        key = (self.post_quant_conv.weight.data_ptr(), self.post_quant_conv.weight._version)
        if self._prep is not None and self._prep_key == key:
            return self._prep
        dev = self.device
        W: Dict[str, torch.Tensor] = {}

        def conv3(name: str, m: _CausalConv3d, cout_multiple: int = 8) -> None:
            w = m.conv.weight.detach().to(dev, F16)                           # [Cout, Cin, 3, 3, 3]
            cout, cin = w.shape[:2]
            cinp, coutp = _pad_to(cin, 64), _pad_to(cout, cout_multiple)
            wp = torch.zeros(coutp, 27, cinp, dtype=F16, device=dev)
            wp[:cout, :, :cin] = w.permute(0, 2, 3, 4, 1).reshape(cout, 27, cin)
            bp = torch.zeros(coutp, dtype=F16, device=dev)
            bp[:cout] = m.conv.bias.detach().to(dev, F16)
            W[name + ".w"], W[name + ".b"] = wp.reshape(coutp, 27 * cinp).contiguous(), bp

        def norm(name: str, m: nn.GroupNorm) -> None:
            W[name + ".g"] = m.weight.detach().to(dev, F16).contiguous()
            W[name + ".b"] = m.bias.detach().to(dev, F16).contiguous()

        def resnet(name: str, m: _Resnet) -> None:
            norm(name + ".norm1", m.norm1)
            conv3(name + ".conv1", m.conv1)
            norm(name + ".norm2", m.norm2)
            conv3(name + ".conv2", m.conv2)
            if hasattr(m, "conv_shortcut"):
                w = m.conv_shortcut.conv.weight.detach().to(dev, F16)
                W[name + ".sc.w"] = w.reshape(w.shape[0], w.shape[1]).contiguous()
                W[name + ".sc.b"] = m.conv_shortcut.conv.bias.detach().to(dev, F16).contiguous()

        d = self.decoder
        lat = self.config.latent_channels
        # post_quant_conv as a [64, 64] GEMM whose zero-padded output feeds conv_in (Cin padded 16 -> 64)
        latp = _pad_to(lat, 64)
        wpq = torch.zeros(latp, latp, dtype=F16, device=dev)
        wpq[:lat, :lat] = self.post_quant_conv.weight.detach().to(dev, F16).reshape(lat, lat)
        bpq = torch.zeros(latp, dtype=F16, device=dev)
        bpq[:lat] = self.post_quant_conv.bias.detach().to(dev, F16)
        W["pq.w"], W["pq.b"] = wpq, bpq
        def attention(name: str, att: Optional[_Attention]) -> None:
            if att is None:
                return
            c = att.to_q.weight.shape[0]
            norm(name + ".gn", att.group_norm)
            scale = c ** -0.5                      # one head of c channels; folded into the q projection
            wq = (att.to_q.weight.detach().float() * scale).to(dev, F16)
            bq = (att.to_q.bias.detach().float() * scale).to(dev, F16)
            W[name + ".qkv.w"] = torch.cat([wq, att.to_k.weight.detach().to(dev, F16), att.to_v.weight.detach().to(dev, F16)]).contiguous()
            W[name + ".qkv.b"] = torch.cat([bq, att.to_k.bias.detach().to(dev, F16), att.to_v.bias.detach().to(dev, F16)]).contiguous()
            W[name + ".o.w"] = att.to_out[0].weight.detach().to(dev, F16).contiguous()
            W[name + ".o.b"] = att.to_out[0].bias.detach().to(dev, F16).contiguous()

        conv3("conv_in", d.conv_in)
        resnet("mid.0", d.mid_block.resnets[0])
        resnet("mid.1", d.mid_block.resnets[1])
        attention("attn", d.mid_block.attentions[0])
        if self._encoder_loaded:
            e = self.encoder
            conv3("enc.conv_in", e.conv_in)
            for i, db in enumerate(e.down_blocks):
                for j, r in enumerate(db.resnets):
                    resnet(f"enc.down.{i}.{j}", r)
                if db.stride is not None:
                    conv3(f"enc.down.{i}.ds", db.downsamplers[0].conv)
            resnet("enc.mid.0", e.mid_block.resnets[0])
            resnet("enc.mid.1", e.mid_block.resnets[1])
            attention("enc.attn", e.mid_block.attentions[0])
            norm("enc.norm_out", e.conv_norm_out)
            conv3("enc.conv_out", e.conv_out, cout_multiple=64)           # zero channels up to quant_conv's K tile
            m2 = 2 * lat
            m2p = _pad_to(m2, 64)
            wq = torch.zeros(_pad_to(m2, 8), m2p, dtype=F16, device=dev)
            wq[:m2, :m2] = self.quant_conv.weight.detach().to(dev, F16).reshape(m2, m2)
            bq = torch.zeros(_pad_to(m2, 8), dtype=F16, device=dev)
            bq[:m2] = self.quant_conv.bias.detach().to(dev, F16)
            W["q.w"], W["q.b"] = wq, bq
        for i, ub in enumerate(d.up_blocks):
            for j, r in enumerate(ub.resnets):
                resnet(f"up.{i}.{j}", r)
            if ub.factor is not None:
                conv3(f"up.{i}.us", ub.upsamplers[0].conv)
        norm("norm_out", d.conv_norm_out)
        conv3("conv_out", d.conv_out)
        self._prep, self._prep_key = W, key
        return W

    # ---- the decoder on channels-last fp16 volumes ------------------------------------------------------------
    def _resnet(self, W, name: str, h: torch.Tensor) -> torch.Tensor:
        g = self.config.norm_num_groups
        a = ops.groupnorm(h, W[name + ".norm1.g"], W[name + ".norm1.b"], g)
        c1 = ops.conv3d_causal(a, W[name + ".conv1.w"], W[name + ".conv1.b"])
        del a
        ops.groupnorm(c1, W[name + ".norm2.g"], W[name + ".norm2.b"], g, out=c1)
        if (name + ".sc.w") in W:
            cin = h.shape[-1]
            res = ops.gemm(h.reshape(-1, cin), W[name + ".sc.w"], W[name + ".sc.b"]).reshape(*h.shape[:3], -1)
        else:
            res = h
        return ops.conv3d_causal(c1, W[name + ".conv2.w"], W[name + ".conv2.b"], resid=res)

    def _attention(self, W, h: torch.Tensor, name: str = "attn") -> torch.Tensor:
        T, H, Wd, C = h.shape
        S = T * H * Wd
        Sp = _pad_to(S, 64)
        x = h.reshape(S, C)
        hn = ops.groupnorm(h, W[name + ".gn.g"], W[name + ".gn.b"], self.config.norm_num_groups, silu=False).reshape(S, C)
        qkv = torch.zeros(Sp, 3 * C, dtype=F16, device=h.device)              # rows >= S stay zero
        ops.gemm(hn, W[name + ".qkv.w"], W[name + ".qkv.b"], out=qkv[:S])
        vt = qkv[:, 2 * C:].t().contiguous()                                   # [C, Sp]; padding columns are zero
        o = ops.attention_framecausal(qkv[:S, :C], qkv[:, C:2 * C], vt, H * Wd, H * Wd)   # q is pre-scaled (_prepare)
        out = ops.gemm(o, W[name + ".o.w"], W[name + ".o.b"], epilogue=ops.EPI_RESIDUAL, resid=x)
        return out.reshape(T, H, Wd, C)

    # ---- the same decoder on a band of rows per rank (enable_tile_parallel) --------------------------------------
    GN_EPS = 1e-6

    def _gn_b(self, B: _Bands, W, name: str, x: torch.Tensor, silu: bool = True, out: Optional[torch.Tensor] = None):
        g = self.config.norm_num_groups
        st = B.global_stats(ops.groupnorm_stats(x, g, self.GN_EPS), x.numel() // g, self.GN_EPS)
        return ops.groupnorm_apply(x, st, W[name + ".g"], W[name + ".b"], g, silu, out)

    def _conv_b(self, B: _Bands, W, name: str, x: torch.Tensor, up=(1, 1, 1), resid: Optional[torch.Tensor] = None):
        """A 3x3x3 convolution over this rank's band with the neighbours' border rows, the exchange overlapped: the
        border rows are posted, the convolution runs over the band as it is (its first and last output rows, computed
        against the kernel's clamp where a neighbour's row belongs, are provisional), and when the rows have arrived the
        `up_h` output rows next to each neighbour are recomputed from a slab of three input rows (neighbour's row, border
        row, the row behind it) and written over the provisional ones.  Same kernel, same operands per output voxel as
        the convolution over the extended band (FRAMEPACK_VAE_HALO_OVERLAP=0 runs that one: exchange, then compute)."""
        w, b = W[name + ".w"], W[name + ".b"]
        if not (B.top or B.bot):
            return ops.conv3d_causal(x, w, b, up=up, resid=resid)
        if not _HALO_OVERLAP:
            xe = B.extend(x)
            re = None
            if resid is not None:                # the skip is added in the kernel's epilogue, as on one GPU: give it
                re = torch.zeros((resid.shape[0], xe.shape[1] * up[1], resid.shape[2], resid.shape[3]), dtype=F16,
                                 device=resid.device)      # the extended shape (halo rows: zeros, they are cut off)
                r0 = up[1] if B.top else 0
                re[:, r0:r0 + resid.shape[1]] = resid
            return B.crop(ops.conv3d_causal(xe, w, b, up=up, resid=re), up[1])
        halo = B.exchange_start(x)
        y = ops.conv3d_causal(x, w, b, up=up, resid=resid)         # runs while the border rows travel
        fu, fd = halo.wait()
        u, Hb = up[1], x.shape[1]

        def fix(rows, mid: int, y0: int):
            """Recompute the output rows of slab row `mid` and write them to y[:, y0 : y0 + u]."""
            slab = torch.cat(rows, dim=1)
            rs = None
            if resid is not None:
                rs = torch.zeros((resid.shape[0], slab.shape[1] * u, resid.shape[2], resid.shape[3]), dtype=F16,
                                 device=resid.device)
                rs[:, mid * u:(mid + 1) * u] = resid[:, y0:y0 + u]
            ys = ops.conv3d_causal(slab, w, b, up=up, resid=rs)
            y[:, y0:y0 + u] = ys[:, mid * u:(mid + 1) * u]
        if Hb == 1:                              # one row: both neighbours (where there are any) in one slab
            fix(([fu] if B.top else []) + [x] + ([fd] if B.bot else []), 1 if B.top else 0, 0)
            return y
        if B.top:
            fix([fu, x[:, :2]], 1, 0)
        if B.bot:
            fix([x[:, -2:], fd], 1, (Hb - 1) * u)
        return y

    def _resnet_b(self, B: _Bands, W, name: str, h: torch.Tensor) -> torch.Tensor:
        a = self._gn_b(B, W, name + ".norm1", h)
        c1 = self._conv_b(B, W, name + ".conv1", a)
        del a
        self._gn_b(B, W, name + ".norm2", c1, out=c1)
        if (name + ".sc.w") in W:
            res = ops.gemm(h.reshape(-1, h.shape[-1]), W[name + ".sc.w"], W[name + ".sc.b"]).reshape(*h.shape[:3], -1)
        else:
            res = h
        return self._conv_b(B, W, name + ".conv2", c1, resid=res)

    def _attention_b(self, B: _Bands, W, h: torch.Tensor, H: int, name: str = "attn") -> torch.Tensor:
        """Mid-block attention with the queries of this rank's band against all keys (frame-causal: a token sees the
        tokens of frames <= its own).  k | v of every band are gathered into (t, h, w) order."""
        T, Hb, Wd, C = h.shape
        Sl, S = T * Hb * Wd, T * H * Wd
        Sp = _pad_to(S, 64)
        x = h.reshape(Sl, C)
        hn = self._gn_b(B, W, name + ".gn", h, silu=False).reshape(Sl, C)
        qkv = ops.gemm(hn, W[name + ".qkv.w"], W[name + ".qkv.b"])
        kv = B.gather_rows(qkv[:, C:].reshape(T, Hb, Wd, 2 * C), 1).reshape(S, 2 * C)
        k_all = torch.zeros(Sp, C, dtype=F16, device=h.device)                # rows >= S stay zero
        k_all[:S] = kv[:, :C]
        vt = torch.zeros(C, Sp, dtype=F16, device=h.device)
        vt[:, :S] = kv[:, C:].t()
        # rows of local frame t see the first (t + 1) H W keys
        o = ops.attention_framecausal(qkv[:, :C], k_all, vt, Hb * Wd, H * Wd)
        out = ops.gemm(o, W[name + ".o.w"], W[name + ".o.b"], epilogue=ops.EPI_RESIDUAL, resid=x)
        return out.reshape(T, Hb, Wd, C)

    def _decode_tile_bands(self, z: torch.Tensor) -> torch.Tensor:
        """`_decode_tile` with H split into one band of rows per rank; every rank returns the whole frames."""
        W = self._prepare()
        cfg = self.config
        lat = cfg.latent_channels
        _, _, T, H, Wd = z.shape
        if not hasattr(self, "_halo_bufs"):
            self._halo_bufs = {}
        B = _Bands(self._tp_group, H, self._halo_bufs)
        x = torch.zeros(T, B.hi - B.lo, Wd, W["pq.w"].shape[0], dtype=F16, device=z.device)
        x[..., :lat] = z[0, :, :, B.lo:B.hi].permute(1, 2, 3, 0).to(F16)
        x = ops.gemm(x.reshape(-1, x.shape[-1]), W["pq.w"], W["pq.b"]).reshape(x.shape)
        h = self._conv_b(B, W, "conv_in", x)
        h = self._resnet_b(B, W, "mid.0", h)
        if cfg.mid_block_add_attention:
            h = self._attention_b(B, W, h, H)
        h = self._resnet_b(B, W, "mid.1", h)
        scale = 1
        for i, ub in enumerate(self.decoder.up_blocks):
            for j in range(len(ub.resnets)):
                h = self._resnet_b(B, W, f"up.{i}.{j}", h)
            if ub.factor is not None:
                h = self._conv_b(B, W, f"up.{i}.us", h, up=ub.factor)
                scale *= ub.factor[1]
        self._gn_b(B, W, "norm_out", h, out=h)
        y = self._conv_b(B, W, "conv_out", h)                                  # [To, scale * Hb, Wo, 8]
        y = B.gather_rows(y[..., :cfg.out_channels].contiguous(), scale)
        return y.permute(3, 0, 1, 2).unsqueeze(0)

    def _decode_tile(self, z: torch.Tensor) -> torch.Tensor:
        """z: [1, latent, T, H, W] on the GPU -> [1, 3, 4(T-1)+1, 8H, 8W] fp16."""
        if getattr(self, "_tp_group", None) is not None:
            return self._decode_tile_bands(z)
        W = self._prepare()
        cfg = self.config
        lat = cfg.latent_channels
        _, _, T, H, Wd = z.shape
        x = torch.zeros(T, H, Wd, W["pq.w"].shape[0], dtype=F16, device=z.device)
        x[..., :lat] = z[0].permute(1, 2, 3, 0).to(F16)
        x = ops.gemm(x.reshape(-1, x.shape[-1]), W["pq.w"], W["pq.b"]).reshape(x.shape)
        h = ops.conv3d_causal(x, W["conv_in.w"], W["conv_in.b"])
        h = self._resnet(W, "mid.0", h)
        if cfg.mid_block_add_attention:
            h = self._attention(W, h)
        h = self._resnet(W, "mid.1", h)
        for i, ub in enumerate(self.decoder.up_blocks):
            for j in range(len(ub.resnets)):
                h = self._resnet(W, f"up.{i}.{j}", h)
            if ub.factor is not None:
                h = ops.conv3d_causal(h, W[f"up.{i}.us.w"], W[f"up.{i}.us.b"], up=ub.factor)
        ops.groupnorm(h, W["norm_out.g"], W["norm_out.b"], cfg.norm_num_groups, out=h)
        y = ops.conv3d_causal(h, W["conv_out.w"], W["conv_out.b"])            # [To, Ho, Wo, 8]; channels >= 3 are padding
        return y[..., :cfg.out_channels].permute(3, 0, 1, 2).unsqueeze(0)

    def _encode_tile(self, x: torch.Tensor) -> torch.Tensor:
        """x: [1, 3, T, H, W] on the GPU -> moments [1, 2*latent, To, H/8, W/8] fp16 (mean | logvar)."""
        W = self._prepare()
        cfg = self.config
        _, cin, T, H, Wd = x.shape
        v = torch.zeros(T, H, Wd, _pad_to(cin, 64), dtype=F16, device=x.device)
        v[..., :cin] = x[0].permute(1, 2, 3, 0).to(F16)
        h = ops.conv3d_causal(v, W["enc.conv_in.w"], W["enc.conv_in.b"])
        del v
        for i, db in enumerate(self.encoder.down_blocks):
            for j in range(len(db.resnets)):
                h = self._resnet(W, f"enc.down.{i}.{j}", h)
            if db.stride is not None:
                h = ops.conv3d_causal_strided(h, W[f"enc.down.{i}.ds.w"], W[f"enc.down.{i}.ds.b"], db.stride)
        h = self._resnet(W, "enc.mid.0", h)
        if cfg.mid_block_add_attention:
            h = self._attention(W, h, "enc.attn")
        h = self._resnet(W, "enc.mid.1", h)
        ops.groupnorm(h, W["enc.norm_out.g"], W["enc.norm_out.b"], cfg.norm_num_groups, out=h)
        y = ops.conv3d_causal(h, W["enc.conv_out.w"], W["enc.conv_out.b"])     # [To, Ho, Wo, 64]; channels >= 32 are zero
        m = ops.gemm(y.reshape(-1, y.shape[-1]), W["q.w"], W["q.b"]).reshape(*y.shape[:3], -1)
        return m[..., :2 * cfg.latent_channels].permute(3, 0, 1, 2).unsqueeze(0)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """`vae.encode(image).latent_dist`.  Inputs of at most `tile_sample_min_num_frames` frames (the reference
        encodes single images); diffusers' framewise encoding of longer clips is not built."""
        if not x.is_cuda:
            raise RuntimeError("framepack_amd VAE encode runs on the GPU only; there is no CPU fallback")
        if not self._encoder_loaded:
            raise RuntimeError("this VAE was loaded from a decoder-only state dict: no encoder weights")
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != self.config.in_channels:
            raise ValueError(f"expected an image volume [1, {self.config.in_channels}, T, H, W], got {tuple(x.shape)}")
        if x.shape[2] > self.tile_sample_min_num_frames:
            raise ValueError(f"{x.shape[2]} frames: framewise encoding of clips longer than "
                             f"{self.tile_sample_min_num_frames} frames is not built")
        sr = self.config.spatial_compression_ratio
        if x.shape[3] % sr or x.shape[4] % sr:
            raise ValueError(f"height and width must be multiples of {sr}, got {tuple(x.shape[3:])}")
        moments = self._encode_tile(x)
        moments = moments.to(self.dtype) if moments.dtype != self.dtype else moments
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    @staticmethod
    def _blend_t(a: torch.Tensor, b: torch.Tensor, extent: int) -> torch.Tensor:
        extent = min(a.shape[2], b.shape[2], extent)
        b = b.clone()
        for x in range(extent):
            b[:, :, x] = a[:, :, -extent + x] * (1 - x / extent) + b[:, :, x] * (x / extent)
        return b

    def enable_window_parallel(self, group=None) -> None:
        """Multi-GPU decode (new design; the reference is single-GPU): the framewise windows are independent
        given the replicated latents (each window re-reads its 4+1 latent frames, the causal convolutions need
        nothing to the left of the window's first frame by construction of diffusers' windows), so window n is
        decoded by rank n % P and broadcast; every rank then does the same cross-fade.  Results are identical to
        the single-GPU framewise decode."""
        import torch.distributed as dist
        if getattr(self, "_tp_group", None) is not None:
            # (a window's owner would wait in its tile's halo exchange for ranks that wait in the window's broadcast)
            raise RuntimeError("enable_window_parallel and enable_tile_parallel exclude each other: a decode is split "
                               "either by windows or by bands of rows")
        self._wp_group = group if group is not None else dist.group.WORLD

    def enable_tile_parallel(self, group=None) -> None:
        """Multi-GPU decode by spatial tiles with halo exchange (new design; the reference is single-GPU, BASELINE.json
        north_star): every decode window is split into one band of rows per rank (`_Bands`), border rows are exchanged
        before every convolution, GroupNorm statistics are all-reduced, the mid-block's keys and values all-gathered.
        Unlike `enable_window_parallel` (whole windows dealt to ranks: at most one rank per window busy, 3 windows in a
        33-frame section) every rank works on every window, and the result equals the single-GPU decode up to the
        rounding of the GroupNorm moments.  Every rank calls decode() with the same latents and gets the same frames."""
        import torch.distributed as dist
        if getattr(self, "_wp_group", None) is not None:
            raise RuntimeError("enable_window_parallel and enable_tile_parallel exclude each other: a decode is split "
                               "either by windows or by bands of rows")
        self._tp_group = group if group is not None else dist.group.WORLD

    def enable_hybrid_parallel(self, window_groups: int) -> None:
        """Windows AND bands (BASELINE config 5: 11 windows of a 5 s 720p clip on 8 GPUs): the world's P ranks form
        `window_groups` groups of P / window_groups ranks; window n is decoded by group n % window_groups, each of its
        ranks on one band of rows (`enable_tile_parallel` inside the group), and then handed to the other groups (rank
        b of a group broadcasts to rank b of every other group: `enable_window_parallel` across the groups).  Where a
        frame has few rows per rank (90 latent rows over 8 ranks) or a clip few windows per rank, the product of the
        two fits better than either.  Every rank calls decode() with the same latents and gets the same frames.
        Collective: every rank of the world must call it (it creates the sub-groups)."""
        import torch.distributed as dist
        P, r = dist.get_world_size(), dist.get_rank()
        if window_groups < 1 or P % window_groups:
            raise ValueError(f"{P} ranks do not split into {window_groups} window groups")
        nb = P // window_groups
        band = [dist.new_group([g * nb + b for b in range(nb)]) for g in range(window_groups)]
        wins = [dist.new_group([g * nb + b for g in range(window_groups)]) for b in range(nb)]
        self._tp_group, self._wp_group = band[r // nb], wins[r % nb]

    def _framewise(self, z: torch.Tensor, tile_fn) -> torch.Tensor:
        """diffusers' _temporal_tiled_decode: windows of t_min+1 latent frames at stride t_stride, the first
        decoded frame of every later window dropped, consecutive windows cross-faded over `blend` frames."""
        tr = self.config.temporal_compression_ratio
        T = z.shape[2]
        t_min, t_stride = self.tile_sample_min_num_frames // tr, self.tile_sample_stride_num_frames // tr
        starts = list(range(0, T, t_stride))
        group = getattr(self, "_wp_group", None)
        rows = []
        if group is None:
            for i in starts:
                d = tile_fn(z[:, :, i:i + t_min + 1])
                rows.append(d[:, :, 1:] if i > 0 else d)
        else:
            import torch.distributed as dist
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            sr = self.config.spatial_compression_ratio
            mine = {n: tile_fn(z[:, :, i:i + t_min + 1]) for n, i in enumerate(starts) if n % world == rank}
            like = next(iter(mine.values())) if mine else None
            for n, i in enumerate(starts):
                n_lat = min(t_min + 1, T - i)
                shape = (1, self.config.out_channels, (n_lat - 1) * tr + 1, z.shape[3] * sr, z.shape[4] * sr)
                if n % world == rank:
                    d = mine[n].contiguous()
                    assert tuple(d.shape) == shape
                else:
                    d = torch.empty(shape, dtype=like.dtype if like is not None else self._wp_dtype(z),
                                    device=z.device)
                dist.broadcast(d, src=dist.get_global_rank(group, n % world), group=group)
                rows.append(d[:, :, 1:] if i > 0 else d)
        blend = self.tile_sample_min_num_frames - self.tile_sample_stride_num_frames
        out = []
        for n, tile in enumerate(rows):
            if n > 0:
                tile = self._blend_t(rows[n - 1], tile, blend)
                out.append(tile[:, :, :self.tile_sample_stride_num_frames])
            else:
                out.append(tile[:, :, :self.tile_sample_stride_num_frames + 1])
        return torch.cat(out, dim=2)[:, :, :(T - 1) * tr + 1]

    def _wp_dtype(self, z: torch.Tensor) -> torch.dtype:
        return F16 if z.is_cuda else z.dtype

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """`vae.decode(latents).sample`.  More than 4 latent frames are decoded in windows of 4+1 latent frames
        at stride 3 and cross-faded over 4 output frames (diffusers' framewise decoding, on by default)."""
        if not z.is_cuda:
            raise RuntimeError("framepack_amd VAE decode runs on the GPU only; there is no CPU fallback")
        if z.dim() != 5 or z.shape[0] != 1:
            raise ValueError(f"expected latents [1, C, T, H, W], got {tuple(z.shape)}")
        t_min = self.tile_sample_min_num_frames // self.config.temporal_compression_ratio
        if not self.use_framewise_decoding or z.shape[2] <= t_min:
            dec = self._decode_tile(z)
        else:
            dec = self._framewise(z, self._decode_tile)
        dec = dec.to(self.dtype) if dec.dtype != self.dtype else dec
        return DecoderOutput(sample=dec) if return_dict else (dec,)


@torch.no_grad()
def vae_decode(latents: torch.Tensor, vae: AutoencoderKLHunyuanVideo, image_mode: bool = False) -> torch.Tensor:
    """ref: diffusers_helper/hunyuan.py:93-103 (same name, arguments and branches)."""
    latents = latents / vae.config.scaling_factor
    if not image_mode:
        return vae.decode(latents.to(device=vae.device, dtype=vae.dtype)).sample
    frames = latents.to(device=vae.device, dtype=vae.dtype).unbind(2)
    return torch.cat([vae.decode(l.unsqueeze(2)).sample for l in frames], dim=2)


@torch.no_grad()
def vae_encode(image: torch.Tensor, vae: AutoencoderKLHunyuanVideo) -> torch.Tensor:
    """ref: diffusers_helper/hunyuan.py:107-110 (same name and arguments)."""
    latents = vae.encode(image.to(device=vae.device, dtype=vae.dtype)).latent_dist.sample()
    return latents * vae.config.scaling_factor


# latent -> RGB preview factors, ref: diffusers_helper/hunyuan.py:62-81 (data of the reference, "From comfyui")
_LATENT_RGB_FACTORS = [
    [-0.0395, -0.0331, 0.0445], [0.0696, 0.0795, 0.0518], [0.0135, -0.0945, -0.0282], [0.0108, -0.0250, -0.0765],
    [-0.0209, 0.0032, 0.0224], [-0.0804, -0.0254, -0.0639], [-0.0991, 0.0271, -0.0669], [-0.0646, -0.0422, -0.0400],
    [-0.0696, -0.0595, -0.0894], [-0.0799, -0.0208, -0.0375], [0.1166, 0.1627, 0.0962], [0.1165, 0.0432, 0.0407],
    [-0.2315, -0.1920, -0.1355], [-0.0270, 0.0401, -0.0821], [-0.0616, -0.0997, -0.0727], [0.0249, -0.0469, -0.1703],
]
_LATENT_RGB_BIAS = [0.0259, -0.0192, -0.0761]


@torch.no_grad()
def vae_decode_fake(latents: torch.Tensor) -> torch.Tensor:
    """ref: diffusers_helper/hunyuan.py:61-89: the per-step preview, a fixed 16 -> 3 projection clamped to [0, 1]
    (K12 of SURVEY.md 2b: trivial, stays a torch op)."""
    weight = torch.tensor(_LATENT_RGB_FACTORS, device=latents.device, dtype=latents.dtype).transpose(0, 1)[:, :, None, None, None]
    bias = torch.tensor(_LATENT_RGB_BIAS, device=latents.device, dtype=latents.dtype)
    return torch.nn.functional.conv3d(latents, weight, bias=bias).clamp(0.0, 1.0)
