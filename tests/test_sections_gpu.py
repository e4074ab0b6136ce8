"""End-to-end composition on the GPU: the section driver running the HIP DiT (through sample_hunyuan) and the HIP
VAE (through vae_decode) for a tiny random-init model: two sections of two UniPC steps.  Checks the contracts
between the pieces (shapes, dtypes, finiteness, determinism for a fixed seed); numerical parity of each piece is
the subject of test_model_gpu.py / test_vae_gpu.py / test_sampler_cpu.py."""
import functools

import pytest
import torch

pytestmark = pytest.mark.gpu

from framepack_amd import sampler, sections, synthetic  # noqa: E402
from framepack_amd.model import HunyuanVideoTransformer3DModelPacked  # noqa: E402
from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _run(seed):
    m = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY).to(BF)
    synthetic.init_synthetic_(dict(m.state_dict()), seed=0)
    m = m.to(DEV).eval()
    vcfg = dict(block_out_channels=(64, 64, 128, 128), norm_num_groups=16)
    vae = AutoencoderKLHunyuanVideo(**vcfg)
    vae.load_state_dict({k: v.half() for k, v in VO.random_state_dict(VO.VAEConfig(**vcfg)).items()})
    vae = vae.to(DEV, torch.float16).eval()
    kw = synthetic.make_inputs(seed=3, frames_latent=3, height=64, width=96, n_text=19, dtype=BF)
    cond = dict(prompt_embeds=kw["encoder_hidden_states"].to(DEV), prompt_embeds_mask=kw["encoder_attention_mask"].to(DEV),
                prompt_poolers=kw["pooled_projections"].to(DEV), image_embeddings=kw["image_embeddings"].to(DEV),
                negative_prompt_embeds=kw["encoder_hidden_states"].to(DEV), negative_prompt_embeds_mask=kw["encoder_attention_mask"].to(DEV),
                negative_prompt_poolers=kw["pooled_projections"].to(DEV), has_negative_prompt=False,
                num_inference_steps=2, real_guidance_scale=1.0, distilled_guidance_scale=10.0, guidance_rescale=0.0,
                device=torch.device(DEV), dtype=BF)
    sample_fn = functools.partial(sampler.sample_hunyuan, transformer=m, sampler="unipc")
    start = torch.randn(1, 16, 1, 8, 12, generator=torch.Generator().manual_seed(1))
    with torch.inference_mode():
        return sections.generate_sections(sample_fn, lambda lat: vae_decode(lat, vae), start, 64, 96, 2,
                                          latent_window_size=3, seed=seed, sample_kwargs=cond)


def test_two_sections_end_to_end():
    res = _run(seed=31337)
    assert [s["total_latent_frames"] for s in res.sections] == [3, 7]
    assert tuple(res.history_latents.shape) == (1, 16, 7 + 19, 8, 12)
    assert tuple(res.history_pixels.shape) == (1, 3, 25, 64, 96) and res.history_pixels.dtype == torch.float16
    assert torch.isfinite(res.history_latents).all() and torch.isfinite(res.history_pixels.float()).all()
    again = _run(seed=31337)
    assert torch.equal(again.history_latents, res.history_latents) and torch.equal(again.history_pixels, res.history_pixels)
    other = _run(seed=1)
    assert not torch.equal(other.history_latents, res.history_latents)
