"""The sampler that drives the DiT: k-model wrapper, flow-match UniPC, section sampling entry point.

Mirror of the reference's interface (same names, arguments, order of operations):
  fm_wrapper / rescale_noise_cfg      ref diffusers_helper/k_diffusion/wrapper.py:8-71
  FlowMatchUniPC / sample_unipc       ref diffusers_helper/k_diffusion/uni_pc_fm.py:16-173
  sample_hunyuan + flux schedule      ref diffusers_helper/pipelines/k_diffusion_hunyuan.py:9-124

MI355X-first difference: every UniPC coefficient (h, r_k, rho_p, rho_c, B_h, h_phi_1) is a function of the
sigma schedule alone, so it is computed on the HOST from a host copy of `sigmas` (same fp32 torch ops, same
order); the device only sees a handful of AXPYs over the latent.  The reference does the same scalar math on
the device and reads it back (`rks`, `torch.tensor(b)`, two `linalg.solve`), i.e. several host syncs per step.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch


def append_dims(x, target_dims):
    return x[(...,) + (None,) * (target_dims - x.ndim)]


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=1.0):
    if guidance_rescale == 0:
        return noise_cfg
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    return guidance_rescale * (noise_cfg * (std_text / std_cfg)) + (1.0 - guidance_rescale) * noise_cfg


def fm_wrapper(transformer, t_scale=1000.0):
    """x0-prediction wrapper around the transformer (ref wrapper.py:19-71)."""

    def k_model(x, sigma, **extra_args):
        dtype = extra_args["dtype"]
        cfg_scale = extra_args["cfg_scale"]
        cfg_rescale = extra_args["cfg_rescale"]
        concat_latent = extra_args["concat_latent"]
        original_dtype = x.dtype
        sigma = sigma.float()
        x = x.to(dtype)
        timestep = (sigma * t_scale).to(dtype)
        hidden_states = x if concat_latent is None else torch.cat([x, concat_latent.to(x)], dim=1)
        pred_positive = transformer(hidden_states=hidden_states, timestep=timestep, return_dict=False,
                                    **extra_args["positive"])[0].float()
        if extra_args.get("has_negative_prompt", False):
            pred_negative = transformer(hidden_states=hidden_states, timestep=timestep, return_dict=False,
                                        **extra_args["negative"])[0].float()
            pred_cfg = pred_negative + cfg_scale * (pred_positive - pred_negative)
        else:
            # the reference adds a zeros tensor here; 0 + s*(p - 0) is the same value without the two passes
            pred_cfg = cfg_scale * pred_positive if cfg_scale != 1.0 else pred_positive
        pred = rescale_noise_cfg(pred_cfg, pred_positive, guidance_rescale=cfg_rescale)
        x0 = x.float() - pred.float() * append_dims(sigma, x.ndim)
        return x0.to(dtype=original_dtype)

    return k_model


class FlowMatchUniPC:
    def __init__(self, model, extra_args, variant="bh1"):
        self.model = model
        self.variant = variant
        self.extra_args = extra_args
        self.movement_scale = extra_args.get("movement_scale", 0.1)

    def model_fn(self, x, t):
        return self.model(x, t, **self.extra_args)

    @staticmethod
    def coefficients(t_prev: List[torch.Tensor], t: torch.Tensor, order: int, variant: str = "bh1"):
        """Host-side scalars of one update (ref uni_pc_fm.py:31-91).  t_prev, t: 0-dim fp32 CPU tensors."""
        lam0 = -torch.log(t_prev[-1])
        h = -torch.log(t) - lam0
        rks = [((-torch.log(t_prev[-(i + 1)]) - lam0) / h) for i in range(1, order)]
        rks_t = torch.tensor([float(r) for r in rks] + [1.0])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        if variant == "bh1":
            B_h = hh
        elif variant == "bh2":
            B_h = torch.expm1(hh)
        else:
            raise NotImplementedError("Bad variant!")
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= (i + 1)
            h_phi_k = h_phi_k / hh - 1 / fact
        R = torch.stack(R)
        b = torch.tensor([float(v) for v in b])
        if order > 1:
            rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
        else:
            rhos_p = None
        rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        return dict(rks=rks, ratio=t / t_prev[-1], h_phi_1=h_phi_1, B_h=B_h, rhos_p=rhos_p, rhos_c=rhos_c)

    def update_fn(self, x, model_prev_list, t_prev_host, t_host, t_dev, order):
        """One predictor-corrector step (ref uni_pc_fm.py:27-114) with host-side scalars."""
        assert order <= len(model_prev_list)
        c = self.coefficients(t_prev_host, t_host, order, self.variant)
        m0 = model_prev_list[-1]
        D1s = [(model_prev_list[-(i + 1)] - m0) / float(c["rks"][i - 1]) for i in range(1, order)]
        x_t_ = float(c["ratio"]) * x - float(c["h_phi_1"]) * m0
        # the k <= 2 term sums of ref uni_pc_fm.py:93-108 (einsum 'k,bkchw->bchw') as plain weighted sums: a library
        # GEMM for two terms costs a launch of a lazily loaded kernel (tens of ms the first time) per step
        def combine(rhos):
            acc = 0
            for d, r in zip(D1s, rhos):
                acc = acc + d * float(r)
            return acc
        pred_res = combine(c["rhos_p"]) if D1s else 0
        x_t = x_t_ - float(c["B_h"]) * pred_res
        model_t = self.model_fn(x_t, t_dev)
        corr_res = combine(c["rhos_c"][:-1]) if D1s else 0
        x_t = x_t_ - float(c["B_h"]) * (corr_res + float(c["rhos_c"][-1]) * (model_t - m0))
        return x_t, model_t

    def sample(self, x, sigmas, callback=None, disable_pbar=False):
        """ref uni_pc_fm.py:116-168 (incl. the step-5 movement roll and the order ramp)."""
        sig_host = sigmas.detach().float().cpu()          # the only device->host copy of the whole loop
        order = min(3, len(sigmas) - 2)
        model_prev_list, t_prev_host = [], []
        for i in range(len(sigmas) - 1):
            vec_t = sigmas[i].expand(x.shape[0])
            if i == 5:
                f, size = x.shape[2], x.shape[-1]
                for j in range(f):
                    amount = math.sin((j / f) * math.pi * 2) * size * self.movement_scale
                    x[:, :, j, :, :] = torch.roll(x[:, :, j, :, :], int(amount), dims=-1)
            if i == 0:
                model_prev_list = [self.model_fn(x, vec_t)]
                t_prev_host = [sig_host[i]]
            else:
                x, model_x = self.update_fn(x, model_prev_list, t_prev_host, sig_host[i], vec_t,
                                            i if i < order else order)
                model_prev_list.append(model_x)
                t_prev_host.append(sig_host[i])
            model_prev_list = model_prev_list[-order:]
            t_prev_host = t_prev_host[-order:]
            if callback is not None:
                callback({"x": x, "i": i, "denoised": model_prev_list[-1], "movement_scale": self.movement_scale})
        return model_prev_list[-1]


def sample_unipc(model, noise, sigmas, extra_args=None, callback=None, disable=False, variant="bh1"):
    assert variant in ["bh1", "bh2"]
    return FlowMatchUniPC(model, extra_args=extra_args, variant=variant).sample(noise, sigmas=sigmas, callback=callback,
                                                                                disable_pbar=disable)


def flux_time_shift(t, mu=1.15, sigma=1.0):
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def calculate_flux_mu(context_length, x1=256, y1=0.5, x2=4096, y2=1.15, exp_max=7.0):
    k = (y2 - y1) / (x2 - x1)
    b = y1 - k * x1
    return min(k * context_length + b, math.log(exp_max))


def get_flux_sigmas_from_mu(n, mu):
    return flux_time_shift(torch.linspace(1, 0, steps=n + 1), mu=mu)


def repeat_to_batch_size(tensor, batch_size):
    """ref diffusers_helper/utils.py:140-154."""
    if tensor is None:
        return None
    first = tensor.shape[0]
    if first == batch_size:
        return tensor
    if batch_size % first != 0:
        raise ValueError(f"Cannot evenly repeat first dim {first} to match batch_size {batch_size}.")
    return tensor.repeat(batch_size // first, *[1] * (tensor.dim() - 1))


@torch.inference_mode()
def sample_hunyuan(transformer, sampler="unipc", initial_latent=None, concat_latent=None, strength=1.0, width=512,
                   height=512, frames=16, real_guidance_scale=1.0, distilled_guidance_scale=6.0,
                   guidance_rescale=0.0, shift=None, num_inference_steps=25, batch_size=None, generator=None,
                   prompt_embeds=None, prompt_embeds_mask=None, prompt_poolers=None, negative_prompt_embeds=None,
                   negative_prompt_embeds_mask=None, negative_prompt_poolers=None, dtype=torch.bfloat16, device=None,
                   negative_kwargs=None, callback=None, has_negative_prompt=None, movement_scale=0.1, **kwargs):
    """One section of `num_inference_steps` denoising steps (ref k_diffusion_hunyuan.py:27-124)."""
    device = device or transformer.device
    if batch_size is None:
        batch_size = int(prompt_embeds.shape[0])
    latents = torch.randn((batch_size, 16, (frames + 3) // 4, height // 8, width // 8), generator=generator,
                          device=generator.device).to(device=device, dtype=torch.float32)
    B, C, T, H, W = latents.shape
    seq_length = T * H * W // 4
    mu = calculate_flux_mu(seq_length, exp_max=7.0) if shift is None else math.log(shift)
    sigmas = get_flux_sigmas_from_mu(num_inference_steps, mu).to(device)
    # a new section: new clean latents and embeddings.  The model keys its cross-step caches by tensor identity and
    # version; dropping them here also covers GPU tensors edited in place under inference_mode (no version counter)
    if hasattr(transformer, "invalidate_step_invariants"):
        transformer.invalidate_step_invariants()
    k_model = fm_wrapper(transformer)
    if initial_latent is not None:
        sigmas = sigmas * strength
        first_sigma = sigmas[0].to(device=device, dtype=torch.float32)
        initial_latent = initial_latent.to(device=device, dtype=torch.float32)
        latents = initial_latent.float() * (1.0 - first_sigma) + latents.float() * first_sigma
    if concat_latent is not None:
        concat_latent = concat_latent.to(latents)
    distilled_guidance = torch.tensor([distilled_guidance_scale * 1000.0] * batch_size).to(device=device, dtype=dtype)
    rep = lambda t: repeat_to_batch_size(t, batch_size)   # noqa: E731
    sampler_kwargs = dict(
        dtype=dtype, cfg_scale=real_guidance_scale, cfg_rescale=guidance_rescale, concat_latent=rep(concat_latent),
        has_negative_prompt=has_negative_prompt, movement_scale=movement_scale,
        positive=dict(pooled_projections=rep(prompt_poolers), encoder_hidden_states=rep(prompt_embeds),
                      encoder_attention_mask=rep(prompt_embeds_mask), guidance=distilled_guidance, **kwargs),
        negative=dict(pooled_projections=rep(negative_prompt_poolers), encoder_hidden_states=rep(negative_prompt_embeds),
                      encoder_attention_mask=rep(negative_prompt_embeds_mask), guidance=distilled_guidance,
                      **(kwargs if negative_kwargs is None else {**kwargs, **negative_kwargs})))
    if sampler == "unipc":
        return sample_unipc(k_model, latents, sigmas, extra_args=sampler_kwargs, disable=False, callback=callback)
    raise NotImplementedError(f"Sampler {sampler} is not supported.")
