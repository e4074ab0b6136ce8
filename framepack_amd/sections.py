"""Headless section driver: the reference worker's generation loop without the Gradio plumbing.

Mirrors ref: framepack/worker.py:325-350 (history buffer, latent-padding schedule), :430-465 (end-frame latent,
index split), :503-505 (clean latents from the history), :621-664 (sampling call, start-latent prepend, history
update), :703-720 (which latent frames are decoded and how they are cross-faded) and
ref: diffusers_helper/utils.py:64-75 (soft_append_bcthw).  It is pure tensor bookkeeping with exact integer
semantics; the two heavy calls are injected (`sample_fn` = framepack_amd.sampler.sample_hunyuan over the HIP DiT,
`decode_fn` = framepack_amd.vae.vae_decode over the HIP VAE), so the same loop runs on ranks without a UI
(sequence-parallel DiT, window-parallel VAE) and in the CPU tests with stand-ins.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch


def latent_paddings(total_latent_sections: int, one_frame_inference: bool = False) -> List[int]:
    """ref worker.py:337-350.  Sections are generated from the END of the video to its start; the padding is
    the number of latent windows left between the start frame and the section."""
    total_latent_sections = int(max(total_latent_sections, 1))                 # ref :95
    if one_frame_inference:
        return [0]
    if total_latent_sections > 4:
        return [3] + [2] * (total_latent_sections - 3) + [1, 0]
    return list(reversed(range(total_latent_sections)))


def section_indices(latent_padding: int, latent_window_size: int, one_frame_inference: bool = False) -> Dict[str, Optional[torch.Tensor]]:
    """ref worker.py:463-481: frame positions [start | padding | window | 1 post | 2 (2x) | 16 (4x)]."""
    pad = latent_padding * latent_window_size
    idx = torch.arange(0, sum([1, pad, latent_window_size, 1, 2, 16])).unsqueeze(0)
    pre, _, latent, post, two, four = idx.split([1, pad, latent_window_size, 1, 2, 16], dim=1)
    out = dict(latent_indices=latent, clean_latent_indices=torch.cat([pre, post], dim=1),
               clean_latent_2x_indices=two, clean_latent_4x_indices=four)
    if one_frame_inference:
        out["latent_indices"] = latent[:, -1:]
        out["clean_latent_2x_indices"] = None
        out["clean_latent_4x_indices"] = None
    return out


def soft_append_bcthw(history: torch.Tensor, current: torch.Tensor, overlap: int = 0) -> torch.Tensor:
    """ref diffusers_helper/utils.py:64-75: `current` appended to `history` along T, the last `overlap` frames of the
    one cross-faded linearly (weights 1 -> 0, ends included) into the first `overlap` frames of the other.  Written
    into one output buffer; the blend keeps the reference's order of operations (w h + (1 - w) c in history's dtype)
    so the frames are bit-identical (`tests/golden/utils.npz`)."""
    th, tc = history.shape[2], current.shape[2]
    ov = max(int(overlap), 0)
    if ov:
        assert th >= ov, f"History length ({th}) must be >= overlap ({ov})"
        assert tc >= ov, f"Current length ({tc}) must be >= overlap ({ov})"
    shape = list(history.shape)
    shape[2] = th + tc - ov
    out = torch.empty(shape, dtype=history.dtype, device=history.device)
    out[:, :, :th - ov] = history[:, :, :th - ov]
    out[:, :, th:] = current[:, :, ov:]
    if ov:
        w = torch.linspace(1, 0, ov, dtype=history.dtype, device=history.device).view(1, 1, ov, 1, 1)
        out[:, :, th - ov:th] = w * history[:, :, th - ov:] + (1 - w) * current[:, :, :ov]
    return out


def crop_or_pad_yield_mask(x: torch.Tensor, length: int):
    """ref diffusers_helper/utils.py:157-169 (the worker brings every prompt embedding to 512 rows with it,
    worker.py:189-190, 392): x [B, F, C] -> ([B, length, C], bool mask [B, length]); rows past F are zeros with mask False,
    a longer x is cut (a view, like the reference) and the mask is all True."""
    B, F, C = x.shape
    keep = min(F, length)
    mask = torch.zeros((B, length), dtype=torch.bool, device=x.device)
    mask[:, :keep] = True
    if F >= length:
        return x[:, :length, :], mask
    y = x.new_zeros((B, length, C))
    y[:, :F] = x
    return y, mask


# (height, width) buckets of ~640^2 pixels, ref diffusers_helper/bucket_tools.py:2-17 (data of the reference).  There is no
# square one: a square input lands on (640, 608), the later of the two equally near entries.
_BUCKETS_640 = tuple(zip((416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 768, 832, 864, 960),
                         (960, 864, 832, 768, 704, 672, 640, 608, 576, 544, 512, 480, 448, 416)))


def find_nearest_bucket(h: int, w: int, resolution: int = 640) -> Tuple[int, int]:
    """ref diffusers_helper/bucket_tools.py:42-79: the bucket whose aspect is nearest to h : w (cross-product distance
    |h bw - w bh|; of equally near ones the LAST in table order), the table scaled by resolution / 640 and truncated
    (the reference's precomputed 320 table is that same rule)."""
    scale = resolution / 640.0
    best, best_metric = None, float("inf")
    for bh, bw in _BUCKETS_640:
        if resolution != 640:
            bh, bw = int(bh * scale), int(bw * scale)
        metric = abs(h * bw - w * bh)
        if metric <= best_metric:
            best, best_metric = (bh, bw), metric
    return best


def pick_ui_section(i_section: int, keys, total_sections: int):
    """ref worker.py:354-367, 404-418 (get_section_prompt / get_section_latent): the per-section maps are keyed by UI
    section number, counted from the START of the video, while the loop generates from the END: processing step
    i_section is UI section `total_sections - i_section - 1`, and it takes the entry of the largest key <= that number
    (the nearest earlier UI section).  Returns that key, or None when no key qualifies."""
    reversed_index = total_sections - i_section - 1
    valid = [k for k in keys if k <= reversed_index]
    return max(valid) if valid else None


@dataclass
class SectionResult:
    history_latents: torch.Tensor              # [1, 16, generated + 19, h, w] fp32 (CPU), newest section first
    history_pixels: Optional[torch.Tensor]     # [1, 3, frames, H, W]
    total_generated_latent_frames: int
    sections: List[dict] = field(default_factory=list)


@torch.no_grad()
def generate_sections(sample_fn: Callable, decode_fn: Optional[Callable], start_latent: torch.Tensor, height: int,
                      width: int, total_latent_sections: int, latent_window_size: int = 9, seed: int = 31337,
                      end_latent: Optional[torch.Tensor] = None, end_frame_strength: float = 1.0,
                      one_frame_inference: bool = False, section_latents: Optional[Dict[int, torch.Tensor]] = None,
                      sample_kwargs: Optional[dict] = None, on_section: Optional[Callable] = None,
                      section_sample_kwargs: Optional[Dict[int, dict]] = None) -> SectionResult:
    """The reference's section loop (ref worker.py:325-332, 430-720).

    sample_fn(frames=..., generator=..., latent_indices=..., clean_latents=..., ... , **sample_kwargs)
        -> generated latents [1, 16, T, h, w]            (sample_hunyuan's contract)
    decode_fn(latents) -> pixels [1, 3, 4(T-1)+1, 8h, 8w] (vae_decode's contract), or None to skip decoding
    section_latents: optional replacements of the start latent keyed by UI section number (ref get_section_latent,
        worker.py:404-426): a processing step uses the entry of the nearest earlier UI section (`pick_ui_section`).
    section_sample_kwargs: optional per-UI-section overrides of `sample_kwargs`, chosen by the same rule - the place of
        the reference's per-section prompts (ref get_section_prompt, worker.py:354-401: prompt_embeds,
        prompt_embeds_mask, prompt_poolers of another prompt; encoding the text is the caller's job, the text encoders
        are outside this path)."""
    rnd = torch.Generator("cpu").manual_seed(seed)
    num_frames = latent_window_size * 4 - 3
    history_latents = torch.zeros((1, 16, 1 + 2 + 16, height // 8, width // 8), dtype=torch.float32)
    history_pixels = None
    total = 0
    log: List[dict] = []
    paddings = latent_paddings(total_latent_sections, one_frame_inference)
    for i_section, latent_padding in enumerate(paddings):
        is_first, is_last = i_section == 0, latent_padding == 0
        if is_first and end_latent is not None:                                   # ref :450-459
            history_latents[:, :, 0:1] = end_latent * end_frame_strength if end_frame_strength != 1.0 else end_latent
        idx = section_indices(latent_padding, latent_window_size, one_frame_inference)
        key = pick_ui_section(i_section, section_latents.keys(), len(paddings)) if section_latents else None
        current = start_latent if key is None else section_latents[key]
        per_section = dict(sample_kwargs or {})
        pkey = pick_ui_section(i_section, section_sample_kwargs.keys(), len(paddings)) if section_sample_kwargs else None
        if pkey is not None:
            per_section.update(section_sample_kwargs[pkey])
        clean_pre = current.to(history_latents)
        clean_post, clean_2x, clean_4x = history_latents[:, :, :1 + 2 + 16].split([1, 2, 16], dim=2)
        kw = dict(clean_latents=torch.cat([clean_pre, clean_post], dim=2), clean_latents_2x=clean_2x,
                  clean_latents_4x=clean_4x, **idx)
        frames = num_frames
        if one_frame_inference:
            # ref :468-481 clears the 2x / 4x indices AND latents, but :503-505 then takes the latents from the history
            # again: the sampler is handed 2x / 4x latents WITHOUT indices (tests/golden/worker.npz, case one_frame),
            # which the transformer ignores (both must be there, hunyuan_video_packed.py:844, 858).  Same here.
            frames = 1
        generated = sample_fn(width=width, height=height, frames=frames, generator=rnd, **kw, **per_section)
        if is_last and not one_frame_inference:                                   # ref :655-656
            generated = torch.cat([start_latent.to(generated), generated], dim=2)
        total += int(generated.shape[2])
        history_latents = torch.cat([generated.to(history_latents), history_latents], dim=2)
        real = history_latents[:, :, :total]
        if decode_fn is not None:                                                 # ref :710-720
            if history_pixels is None:
                history_pixels = decode_fn(real).cpu()
            else:
                n_lat = (latent_window_size * 2 + 1) if is_last else (latent_window_size * 2)
                history_pixels = soft_append_bcthw(decode_fn(real[:, :, :n_lat]).cpu(), history_pixels,
                                                   latent_window_size * 4 - 3)
        log.append(dict(section=i_section, latent_padding=latent_padding, is_last=is_last, total_latent_frames=total,
                        ui_section_latent=key, ui_section_kwargs=pkey,
                        pixel_frames=None if history_pixels is None else int(history_pixels.shape[2])))
        if on_section is not None:
            on_section(log[-1], history_pixels)
    return SectionResult(history_latents, history_pixels, total, log)
