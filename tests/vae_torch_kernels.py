"""TEST INFRASTRUCTURE: torch stand-ins with the signatures of the VAE entries of framepack_amd.ops, so that the
multi-rank logic of framepack_amd.vae (bands, halo exchange, statistics all-reduce, key/value gather) can run on CPU
tensors over a gloo group.  fp32 arithmetic, fp16 storage like the kernels.  Never imported by the product: the product
path has no CPU kernels (framepack_amd.ops raises for CPU tensors)."""
import torch
import torch.nn.functional as F

F16 = torch.float16
EPI_BIAS, EPI_GELU, EPI_GELU_FROM, EPI_GATE_RESIDUAL, EPI_RESIDUAL = 0, 1, 2, 3, 4


def gemm(a, weight, bias, out=None, epilogue=EPI_BIAS, gelu_from=0, resid=None, gate=None, a2=None):
    assert epilogue in (EPI_BIAS, EPI_RESIDUAL) and a2 is None
    y = a.float() @ weight.float().t()
    if bias is not None:
        y = y + bias.float()
    if epilogue == EPI_RESIDUAL:
        y = y + resid.float()
    y = y.to(a.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def conv3d_causal(x, weight, bias, up=(1, 1, 1), resid=None, out=None):
    """x [T, H, W, Cin] fp16, weight [Cout, 27 * Cin] tap-major: the contract of fp_conv3d_causal_f16."""
    T, H, W, cin = x.shape
    cout = weight.shape[0]
    w5 = weight.float().view(cout, 3, 3, 3, cin).permute(0, 4, 1, 2, 3)
    v = x.float().permute(3, 0, 1, 2).unsqueeze(0)
    if up[0] == 2:                                              # first frame not doubled in time
        v = torch.cat([v[:, :, :1], v[:, :, 1:].repeat_interleave(2, dim=2)], dim=2)
    v = v.repeat_interleave(up[1], dim=3).repeat_interleave(up[2], dim=4)
    v = F.pad(v, (1, 1, 1, 1, 2, 0), mode="replicate")
    y = F.conv3d(v, w5, None if bias is None else bias.float())[0].permute(1, 2, 3, 0)
    if resid is not None:
        y = y + resid.float()
    y = y.to(F16).contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


def groupnorm_stats(x, groups, eps=1e-6, stats=None):
    C = x.shape[-1]
    v = x.reshape(-1, groups, C // groups).double()
    mean = v.mean(dim=(0, 2))
    var = (v * v).mean(dim=(0, 2)) - mean * mean
    st = torch.stack([mean, 1.0 / torch.sqrt(var.clamp_min(0) + eps)], dim=1).float().reshape(-1)
    if stats is not None:
        stats.copy_(st)
        return stats
    return st


def groupnorm_apply(x, stats, gamma, beta, groups, silu=True, out=None):
    C = x.shape[-1]
    st = stats.view(groups, 2)
    mean = st[:, 0].repeat_interleave(C // groups)
    rstd = st[:, 1].repeat_interleave(C // groups)
    y = (x.float() - mean) * rstd * gamma.float() + beta.float()
    if silu:
        y = F.silu(y)
    y = y.to(F16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def groupnorm(x, gamma, beta, groups, eps=1e-6, silu=True, out=None):
    return groupnorm_apply(x, groupnorm_stats(x, groups, eps), gamma, beta, groups, silu, out)


def attention_framecausal(q, k, vt, q_per_frame, k_per_frame, out=None, scale=1.0):
    Sq, C = q.shape
    T = Sq // q_per_frame
    Sk = T * k_per_frame
    s = q.float() @ k[:Sk].float().t() * scale
    valid = (torch.arange(Sq) // q_per_frame + 1) * k_per_frame
    mask = torch.arange(Sk)[None, :] < valid[:, None]
    p = torch.softmax(s.masked_fill(~mask, float("-inf")), dim=-1).to(F16)
    y = (p.float() @ vt[:, :Sk].float().t()).to(F16)
    if out is not None:
        out.copy_(y)
        return out
    return y
