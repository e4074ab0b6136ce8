"""Stand-in for the parts of diffusers==0.33.1 the reference's model file imports.

Used ONLY by tests/golden/make_golden.py, in the build container, to let the reference's own
`diffusers_helper/models/hunyuan_video_packed.py` and `diffusers_helper/dit_common.py` import and
run on CPU (the real package is not installed and cannot be fetched).  The classes below are
restatements of the published diffusers 0.33.1 behaviour from recollection - NOT verifiable here -
so golden vectors made through them pin everything that lives in /root/reference and leave the
diffusers containers "parity unpinned" (DESIGN.md "Oracle").
"""
import math
import sys
import types
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = self.scale * emb
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__()
        out_features = out_features or hidden_size
        self.linear_1 = nn.Linear(in_features, hidden_size)
        assert act_fn == "silu"
        self.act_1 = nn.SiLU()
        self.linear_2 = nn.Linear(hidden_size, out_features)

    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))


class RMSNorm(nn.Module):
    def __init__(self, dim, eps, elementwise_affine=True, bias=False):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None
        self.bias = None

    def forward(self, x):          # replaced by dit_common.RMSNorm_forward at import
        raise NotImplementedError


LayerNorm = nn.LayerNorm           # diffusers aliases nn.LayerNorm on torch >= 2.1


class FP32LayerNorm(nn.LayerNorm):
    pass


class AdaLayerNormContinuous(nn.Module):
    pass


class _GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none"):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class _LinearActivation(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.silu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu"):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim_out or dim
        if activation_fn == "gelu-approximate":
            act = _GELU(dim, inner, approximate="tanh")
        elif activation_fn == "linear-silu":
            act = _LinearActivation(dim, inner)
        else:
            raise ValueError(activation_fn)
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner, dim_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class _AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None):
        B, S, _ = hidden_states.shape
        if attention_mask is not None:
            if attention_mask.shape[0] < B * attn.heads:
                attention_mask = attention_mask.repeat_interleave(attn.heads, dim=0)
            attention_mask = attention_mask.view(B, attn.heads, -1, attention_mask.shape[-1])
        q = attn.to_q(hidden_states).view(B, S, attn.heads, -1).transpose(1, 2)
        k = attn.to_k(hidden_states).view(B, S, attn.heads, -1).transpose(1, 2)
        v = attn.to_v(hidden_states).view(B, S, attn.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, S, -1).to(q.dtype)
        return attn.to_out[1](attn.to_out[0](o))


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, qk_norm=None,
                 added_kv_proj_dim=None, out_dim=None, context_pre_only=None, pre_only=False, eps=1e-5,
                 processor=None):
        super().__init__()
        inner = out_dim if out_dim is not None else dim_head * heads
        self.heads = out_dim // dim_head if out_dim is not None else heads
        out_dim = out_dim if out_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.norm_q = self.norm_k = None
        if qk_norm == "rms_norm":
            self.norm_q = RMSNorm(dim_head, eps=eps)
            self.norm_k = RMSNorm(dim_head, eps=eps)
        if added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_q_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            if qk_norm == "rms_norm":
                self.norm_added_q = RMSNorm(dim_head, eps=eps)
                self.norm_added_k = RMSNorm(dim_head, eps=eps)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, out_dim, bias=True), nn.Dropout(0.0)])
        if context_pre_only is not None and not context_pre_only:
            self.to_add_out = nn.Linear(inner, out_dim, bias=True)
        self.processor = processor if processor is not None else _AttnProcessor2_0()

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        import inspect
        ok = set(inspect.signature(self.processor.__call__).parameters.keys())
        kw = {k: v for k, v in kw.items() if k in ok}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor


class ModelMixin(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class ConfigMixin:
    pass


class PeftAdapterMixin:
    pass


class FromOriginalModelMixin:
    pass


def register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        cfg = {k: p.default for k, p in sig.parameters.items() if k != "self" and p.default is not inspect._empty}
        names = [k for k in sig.parameters if k != "self"]
        cfg.update(dict(zip(names, args)))
        cfg.update(kwargs)
        self.config = dict(cfg)
        init(self, *args, **kwargs)
    return wrapper


class _Logging:
    @staticmethod
    def get_logger(name):
        import logging
        return logging.getLogger(name)


def install():
    """Register the stand-in modules under the names the reference imports."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("diffusers")
    mod("diffusers.loaders", FromOriginalModelMixin=FromOriginalModelMixin, PeftAdapterMixin=PeftAdapterMixin)
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.utils", logging=_Logging)
    mod("diffusers.models")
    mod("diffusers.models.attention", FeedForward=FeedForward)
    mod("diffusers.models.attention_processor", Attention=Attention)
    mod("diffusers.models.embeddings", TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps,
        PixArtAlphaTextProjection=PixArtAlphaTextProjection)
    mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=Transformer2DModelOutput)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.normalization", RMSNorm=RMSNorm, LayerNorm=LayerNorm, FP32LayerNorm=FP32LayerNorm,
        AdaLayerNormContinuous=AdaLayerNormContinuous)
