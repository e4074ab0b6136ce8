"""Cases and stand-in collaborators for the section loop, shared by tests/golden/make_worker_golden.py (which runs the
REFERENCE's framepack/worker.py with them and records what its bookkeeping hands over) and tests/test_sections_cpu.py
(which replays the cases through framepack_amd.sections.generate_sections and compares).  Nothing here is product code."""
import numpy as np
import torch


def cases():
    """name -> arguments of the worker that matter to the bookkeeping.  (No end-frame case: at this commit the reference's
    end-frame branch raises before any section runs - worker.py:228-229 permutes the 5-D tensor a second time - so the
    end-latent path, worker.py:450-459, has no reference behaviour to record.)"""
    return {
        "one_section": dict(image_hw=(100, 100), sections=1, window=9, seed=31337),
        "three_sections_neg_prompt": dict(image_hw=(90, 160), sections=3, window=9, seed=7, n_prompt="blurry"),
        "five_sections_window5": dict(image_hw=(160, 90), sections=5, window=5, seed=11),
        "four_sections_window3": dict(image_hw=(64, 64), sections=4, window=3, seed=3),
        "one_frame": dict(image_hw=(120, 100), sections=2, window=9, seed=5, one_frame=True),
        # UI rows (section number, image?, prompt): images at 0 and 2, prompts at 1 and 2; row 3 is past the end
        "ui_sections": dict(image_hw=(100, 130), sections=4, window=4, seed=13,
                            ui=[(0, True, ""), (1, False, "second prompt"), (2, True, "third prompt"), (7, True, "never")]),
        "full_resolution_table": dict(image_hw=(30, 50), sections=2, window=2, seed=17, resolution_scale="Full (1x)"),
    }


def image(hw, seed):
    g = np.random.default_rng(seed)
    return g.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)


def section_settings(case):
    if "ui" not in case:
        return None
    return [[num, image(case["image_hw"], 10 + num) if has_img else None, prompt] for num, has_img, prompt in case["ui"]]


def sub(t):
    """What is stored of a [B, C, T, H, W] tensor: every frame and channel at a grid of spatial positions (the loop only
    moves and blends whole frames)."""
    return t[..., ::7, ::9].float().contiguous().numpy()


def stub_prompt(prompt):
    """(llama_vec [1, n, 8], clip pooler [1, 4]) as a fixed function of the prompt text."""
    n = 3 + len(prompt) % 5
    v = float(sum(map(ord, prompt)) % 97) / 97.0
    return torch.full((1, n, 8), v), torch.full((1, 4), -v)


def stub_encode_hw(k, h, w):
    """The k-th vae_encode call of a run (start image first, then the UI rows' images in row order) -> [1, 16, 1, h, w]."""
    return torch.randn((1, 16, 1, h, w), generator=torch.Generator().manual_seed(1000 + k))


def stub_decode(latents):
    """latents [1, 16, T, h, w] -> 'pixels' [1, 3, 4 (T - 1) + 1, h, w] (latent-sized: the loop never looks at the
    spatial size), mixing neighbouring latent frames like a causal decoder does."""
    x = latents[:, :3].float() + 0.1 * latents[:, 3:6].float()
    t = x.shape[2]
    up = x.repeat_interleave(4, dim=2)[:, :, 3:]
    if t > 1:
        up = up + 0.05 * torch.roll(up, 1, dims=2)
    return up


def stub_sample(rec, width, height, frames, generator, latent_indices, clean_latents, clean_latent_indices,
                clean_latents_2x, clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices, prompt_embeds,
                prompt_embeds_mask, prompt_poolers, **kw):
    """sample_hunyuan's place: record what the loop hands over, return latents drawn from ITS generator."""
    none = np.zeros((0,), dtype=np.float32)
    rec["samples"].append(dict(
        dims=np.array([width, height, frames], dtype=np.int64),
        latent_indices=latent_indices.numpy(), clean_latent_indices=clean_latent_indices.numpy(),
        clean_latent_2x_indices=none if clean_latent_2x_indices is None else clean_latent_2x_indices.numpy(),
        clean_latent_4x_indices=none if clean_latent_4x_indices is None else clean_latent_4x_indices.numpy(),
        clean_latents=sub(clean_latents),
        clean_latents_2x=none if clean_latents_2x is None else sub(clean_latents_2x),
        clean_latents_4x=none if clean_latents_4x is None else sub(clean_latents_4x),
        prompt=np.array([float(prompt_embeds[0, 0, 0]), float(prompt_embeds_mask.sum()), float(prompt_poolers[0, 0])]),
    ))
    t = (frames + 3) // 4
    x = torch.randn((1, 16, t, height // 8, width // 8), generator=generator)
    return x * 0.5 + 0.01 * clean_latents.float().mean()
