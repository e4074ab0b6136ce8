"""End-to-end composition on the GPU: the section driver running the HIP DiT (through sample_hunyuan) and the HIP
VAE (through vae_decode) for a tiny random-init model: two sections of two UniPC steps - against the same loop
composed from the CPU oracles (oracle/sections_oracle.generate over dit_oracle, sampler_oracle, vae_oracle), and the
contracts between the pieces (shapes, dtypes, finiteness, determinism for a fixed seed)."""
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


VCFG = dict(block_out_channels=(64, 64, 128, 128), norm_num_groups=16)


def _run(seed, keep=None):
    m = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY).to(BF)
    synthetic.init_synthetic_(dict(m.state_dict()), seed=0)
    vae_sd = {k: v.half() for k, v in VO.random_state_dict(VO.VAEConfig(**VCFG)).items()}
    kw = synthetic.make_inputs(seed=3, frames_latent=3, height=64, width=96, n_text=19, dtype=BF)
    if keep is not None:
        keep.update(dit={k: v.float().clone() for k, v in m.state_dict().items()}, vae=vae_sd, kw=kw)
    m = m.to(DEV).eval()
    vae = AutoencoderKLHunyuanVideo(**VCFG)
    vae.load_state_dict(vae_sd)
    vae = vae.to(DEV, torch.float16).eval()
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


def test_two_sections_match_the_loop_composed_from_the_oracles():
    """The whole path - padding schedule, index split, clean latents from the history, noise draw, sigma schedule,
    guidance, k-model, UniPC, DiT, history update, decode windows, cross-fade - on the GPU against the same loop on
    the CPU oracles (fp32 on the bf16-rounded weights).  Two sections x two steps through a 2+2-block DiT: the bf16
    path stays within a few per cent of the fp32 one; a wrong index, window or blend is an O(1) difference."""
    import math
    from oracle import dit_oracle as O
    from oracle import sections_oracle as SECO
    keep = {}
    res = _run(seed=31337, keep=keep)
    cfg = O.DiTConfig(**synthetic.CONFIG_TINY)
    vcfg = VO.VAEConfig(**VCFG)
    kw = keep["kw"]
    Pv = {k: v.float() for k, v in keep["vae"].items()}
    cond = dict(encoder_hidden_states=kw["encoder_hidden_states"].float(), encoder_attention_mask=kw["encoder_attention_mask"],
                pooled_projections=kw["pooled_projections"].float(), image_embeddings=kw["image_embeddings"].float(),
                guidance_scale=10.0)
    start = torch.randn(1, 16, 1, 8, 12, generator=torch.Generator().manual_seed(1))
    hist, pix = SECO.generate(lambda hs, t, **k: O.dit_forward(keep["dit"], cfg, hs, t, **k),
                              lambda lat: VO.vae_decode(Pv, vcfg, lat.half().float()), start, 64, 96, 2, 3, 31337, cond,
                              steps=2)
    assert tuple(hist.shape) == tuple(res.history_latents.shape) and tuple(pix.shape) == tuple(res.history_pixels.shape)
    real = res.history_latents[:, :, :7]
    err = ((real.double() - hist[:, :, :7].double()).norm() / hist[:, :, :7].double().norm()).item()
    mse = ((res.history_pixels.double() - pix.double()) ** 2).mean().item()
    psnr = 10 * math.log10(4.0 / max(mse, 1e-30))
    print(f"sections end to end: latents rel-L2 {err:.4f}, pixels PSNR {psnr:.1f} dB")
    assert err < 2e-2 and psnr >= 40.0                # measured on MI355X: 4.7e-3, 58.5 dB
