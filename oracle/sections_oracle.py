"""CPU restatement of the reference's section bookkeeping (TEST INFRASTRUCTURE, see oracle/__init__.py).

Written independently of framepack_amd/sections.py, from ref: framepack/worker.py:337-350 (padding schedule),
:463-465 (index layout), :655-664 / :703-720 (history and decode windows), diffusers_helper/utils.py:64-75
(cross-fade).  Pinned by tests/golden/worker.npz - what the reference's own worker() handed to its sampler, decoder and
video writer when run with recording stand-ins for those (tests/golden/make_worker_golden.py): padding schedule, index
layout and frame counts here, the whole loop in framepack_amd/sections.py (tests/test_sections_cpu.py)."""
from __future__ import annotations

import numpy as np


def paddings(n_sections: int):
    n = max(int(n_sections), 1)
    return [3] + [2] * (n - 3) + [1, 0] if n > 4 else list(range(n - 1, -1, -1))


def indices(padding: int, window: int):
    p = padding * window
    pos = np.arange(1 + p + window + 1 + 2 + 16)
    cuts = np.cumsum([1, p, window, 1, 2, 16])
    pre, _, lat, post, two, four = np.split(pos, cuts[:-1])
    return dict(latent=lat, clean=np.concatenate([pre, post]), two=two, four=four)


def crossfade(history: np.ndarray, current: np.ndarray, overlap: int) -> np.ndarray:
    """history, current: [frames, ...]; `history` keeps its head, `current` its tail."""
    if overlap <= 0:
        return np.concatenate([history, current])
    w = np.linspace(1, 0, overlap, dtype=history.dtype).reshape(-1, *([1] * (history.ndim - 1)))
    mid = w * history[-overlap:] + (1 - w) * current[:overlap]
    return np.concatenate([history[:-overlap], mid, current[overlap:]])


def expected_frame_counts(n_sections: int, window: int):
    """(latent frames, pixel frames) after every section: each section adds `window` latent frames, the last one
    also the start frame; pixels = 4 * latents - 3."""
    out, lat = [], 0
    pads = paddings(n_sections)
    for i, p in enumerate(pads):
        lat += window + (1 if p == 0 else 0)
        out.append((lat, 4 * lat - 3))
    return out


def generate(dit_forward, vae_decode, start_latent, height: int, width: int, n_sections: int, window: int, seed: int,
             cond: dict, steps: int, end_latent=None, end_strength: float = 1.0):
    """The reference's section loop composed from the OTHER oracles (ref: framepack/worker.py:325-350 setup,
    :430-465 per-section inputs, :503-505 clean latents, :621-664 sampling and history update, :703-720 decode windows
    and cross-fade; diffusers_helper/pipelines/k_diffusion_hunyuan.py:57-124 noise, schedule, guidance).

    dit_forward(hidden_states, timestep, **kwargs) -> prediction      (oracle.dit_oracle.dit_forward with P, cfg bound)
    vae_decode(latents) -> pixels                                      (oracle.vae_oracle.vae_decode with P, cfg bound)
    cond: encoder_hidden_states, encoder_attention_mask, pooled_projections, image_embeddings, guidance_scale.
    Returns (history_latents, history_pixels) as torch tensors, fp32."""
    import torch
    from . import sampler_oracle as SO
    rnd = torch.Generator("cpu").manual_seed(seed)
    h8, w8 = height // 8, width // 8
    history = torch.zeros(1, 16, 1 + 2 + 16, h8, w8)
    pixels, total = None, 0
    pads = paddings(n_sections)
    for i_section, pad in enumerate(pads):
        last = pad == 0
        if i_section == 0 and end_latent is not None:
            history[:, :, 0:1] = end_latent * end_strength
        ix = indices(pad, window)
        as_idx = lambda a: torch.from_numpy(np.asarray(a)).long().unsqueeze(0)      # noqa: E731
        clean = torch.cat([start_latent.float(), history[:, :, :1]], dim=2)
        two, four = history[:, :, 1:3], history[:, :, 3:19]
        frames = window * 4 - 3
        noise = torch.randn((1, 16, (frames + 3) // 4, h8, w8), generator=rnd)
        sig = SO.flux_sigmas(steps, SO.calculate_flux_mu(noise.shape[2] * h8 * w8 // 4, exp_max=7.0))
        guidance = torch.tensor([cond["guidance_scale"] * 1000.0]).to(torch.bfloat16).float()
        positive = dict(encoder_hidden_states=cond["encoder_hidden_states"],
                        encoder_attention_mask=cond["encoder_attention_mask"],
                        pooled_projections=cond["pooled_projections"], guidance=guidance,
                        image_embeddings=cond["image_embeddings"], latent_indices=as_idx(ix["latent"]),
                        clean_latents=clean, clean_latent_indices=as_idx(ix["clean"]), clean_latents_2x=two,
                        clean_latent_2x_indices=as_idx(ix["two"]), clean_latents_4x=four,
                        clean_latent_4x_indices=as_idx(ix["four"]))

        def transformer(hidden_states, timestep, return_dict, **kw):
            return (dit_forward(hidden_states.float(), timestep.float(), **kw),)

        def model_fn(x, t):
            return SO.k_model(transformer, x, t, dtype=torch.bfloat16, cfg_scale=1.0, cfg_rescale=0.0,
                              positive=positive)
        gen = SO.unipc_sample(model_fn, noise.float(), sig)
        if last:
            gen = torch.cat([start_latent.float(), gen], dim=2)
        total += gen.shape[2]
        history = torch.cat([gen, history], dim=2)
        real = history[:, :, :total]
        if pixels is None:
            pixels = vae_decode(real)
        else:
            n_lat = window * 2 + (1 if last else 0)
            cur = vae_decode(real[:, :, :n_lat])
            ov = window * 4 - 3
            pixels = torch.from_numpy(crossfade(cur[0].permute(1, 0, 2, 3).numpy(), pixels[0].permute(1, 0, 2, 3).numpy(),
                                                ov)).permute(1, 0, 2, 3).unsqueeze(0)
    return history, pixels
