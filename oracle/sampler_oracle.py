"""CPU restatement of the sampler that drives the DiT.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `ref:` paths are relative to /root/reference.
Pinned by tests/golden/sampler.npz (made from the reference's own importable modules
diffusers_helper/k_diffusion/{uni_pc_fm,wrapper}.py).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch

Tensor = torch.Tensor


def flux_time_shift(t, mu: float = 1.15, sigma: float = 1.0):
    """ref: diffusers_helper/pipelines/k_diffusion_hunyuan.py:9-10."""
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def calculate_flux_mu(context_length: int, x1=256, y1=0.5, x2=4096, y2=1.15, exp_max=7.0) -> float:
    """ref: k_diffusion_hunyuan.py:13-18."""
    k = (y2 - y1) / (x2 - x1)
    b = y1 - k * x1
    return min(k * context_length + b, math.log(exp_max))


def flux_sigmas(n: int, mu: float) -> Tensor:
    """ref: k_diffusion_hunyuan.py:21-24."""
    return flux_time_shift(torch.linspace(1, 0, steps=n + 1), mu=mu)


def rescale_noise_cfg(noise_cfg: Tensor, noise_pred_text: Tensor, guidance_rescale: float = 1.0) -> Tensor:
    """ref: diffusers_helper/k_diffusion/wrapper.py:8-16."""
    if guidance_rescale == 0:
        return noise_cfg
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1.0 - guidance_rescale) * noise_cfg


def k_model(transformer: Callable, x: Tensor, sigma: Tensor, *, dtype, cfg_scale, cfg_rescale,
            positive: dict, negative: Optional[dict] = None, has_negative_prompt: bool = False,
            concat_latent: Optional[Tensor] = None, t_scale: float = 1000.0) -> Tensor:
    """ref: wrapper.py:20-69 - cast, timestep = sigma*1000, CFG combine, rescale, eps -> x0."""
    original = x.dtype
    sigma = sigma.float()
    x = x.to(dtype)
    timestep = (sigma * t_scale).to(dtype)
    hs = x if concat_latent is None else torch.cat([x, concat_latent.to(x)], dim=1)
    pos = transformer(hidden_states=hs, timestep=timestep, return_dict=False, **positive)[0].float()
    if has_negative_prompt:
        neg = transformer(hidden_states=hs, timestep=timestep, return_dict=False, **negative)[0].float()
    else:
        neg = torch.zeros_like(pos)
    pred = rescale_noise_cfg(neg + cfg_scale * (pos - neg), pos, guidance_rescale=cfg_rescale)
    x0 = x.float() - pred.float() * sigma[(...,) + (None,) * (x.ndim - sigma.ndim)]
    return x0.to(dtype=original)


def _bc(v: Tensor, dims: int) -> Tensor:
    return v[(...,) + (None,) * (dims - 1)]


def unipc_update(model_fn: Callable, x: Tensor, model_prev: List[Tensor], t_prev: List[Tensor], t: Tensor,
                 order: int, variant: str = "bh1"):
    """ref: diffusers_helper/k_diffusion/uni_pc_fm.py:27-114."""
    assert order <= len(model_prev)
    dims = x.dim()
    t0 = t_prev[-1]
    lam0 = -torch.log(t0)
    lam_t = -torch.log(t)
    m0 = model_prev[-1]
    h = lam_t - lam0

    rks, D1s = [], []
    for i in range(1, order):
        lam_i = -torch.log(t_prev[-(i + 1)])
        rk = ((lam_i - lam0) / h)[0]
        rks.append(rk)
        D1s.append((model_prev[-(i + 1)] - m0) / rk)
    rks.append(1.0)
    rks = torch.tensor(rks, device=x.device)

    hh = -h[0]
    h_phi_1 = torch.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    fact = 1
    if variant == "bh1":
        B_h = hh
    elif variant == "bh2":
        B_h = torch.expm1(hh)
    else:
        raise NotImplementedError("Bad variant!")

    R, b = [], []
    for i in range(1, order + 1):
        R.append(torch.pow(rks, i - 1))
        b.append(h_phi_k * fact / B_h)
        fact *= (i + 1)
        h_phi_k = h_phi_k / hh - 1 / fact
    R = torch.stack(R)
    b = torch.tensor(b, device=x.device)

    if len(D1s) > 0:
        D1 = torch.stack(D1s, dim=1)
        rhos_p = torch.tensor([0.5], device=b.device) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
    else:
        D1, rhos_p = None, None
    rhos_c = torch.tensor([0.5], device=b.device) if order == 1 else torch.linalg.solve(R, b)

    x_t_ = _bc(t / t0, dims) * x - _bc(h_phi_1, dims) * m0
    pred_res = torch.tensordot(D1, rhos_p, dims=([1], [0])) if D1 is not None else 0
    x_t = x_t_ - _bc(B_h, dims) * pred_res
    model_t = model_fn(x_t, t)
    corr_res = torch.tensordot(D1, rhos_c[:-1], dims=([1], [0])) if D1 is not None else 0
    x_t = x_t_ - _bc(B_h, dims) * (corr_res + rhos_c[-1] * (model_t - m0))
    return x_t, model_t


def movement_roll_(x: Tensor, movement_scale: float) -> None:
    """ref: uni_pc_fm.py:123-137 - in-place per-frame roll along W, applied at step i == 5."""
    f = x.shape[2]
    size = x.shape[-1]
    for j in range(f):
        amount = math.sin((j / f) * math.pi * 2) * size * movement_scale
        x[:, :, j, :, :] = torch.roll(x[:, :, j, :, :], int(amount), dims=-1)


def unipc_sample(model_fn: Callable, x: Tensor, sigmas: Tensor, movement_scale: float = 0.1,
                 callback: Optional[Callable] = None, variant: str = "bh1") -> Tensor:
    """ref: uni_pc_fm.py:116-168."""
    order = min(3, len(sigmas) - 2)
    model_prev: List[Tensor] = []
    t_prev: List[Tensor] = []
    for i in range(len(sigmas) - 1):
        vec_t = sigmas[i].expand(x.shape[0])
        if i == 5:
            movement_roll_(x, movement_scale)
        if i == 0:
            model_prev = [model_fn(x, vec_t)]
            t_prev = [vec_t]
        else:
            x, mx = unipc_update(model_fn, x, model_prev, t_prev, vec_t, i if i < order else order, variant)
            model_prev.append(mx)
            t_prev.append(vec_t)
        model_prev = model_prev[-order:]
        t_prev = t_prev[-order:]
        if callback is not None:
            callback({"x": x, "i": i, "denoised": model_prev[-1]})
    return model_prev[-1]
