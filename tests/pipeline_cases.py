"""Seeded inputs and a stub transformer for the sampling pipeline (`sample_hunyuan`: noise draw, flux-shift sigma
schedule, CFG argument packing, UniPC).  Shared by tests/golden/make_golden.py - which runs the REFERENCE's
diffusers_helper/pipelines/k_diffusion_hunyuan.py on them and stores the results in tests/golden/pipeline.npz - and by
tests/test_sampler_cpu.py, which runs framepack_amd.sampler.sample_hunyuan on the same."""
import torch


class StubTransformer:
    """A deterministic stand-in for the DiT: its prediction depends on every argument the pipeline packs (text
    embeddings, mask, pooled vector, guidance, the caller's extra kwargs), so a wrong packing changes the result."""
    device = torch.device("cpu")
    dtype = torch.float32

    def __call__(self, hidden_states, timestep, encoder_hidden_states=None, encoder_attention_mask=None,
                 pooled_projections=None, guidance=None, return_dict=False, bias=0.0, latent_indices=None, **kw):
        assert not kw, sorted(kw)
        b = hidden_states.shape[0]
        ctx = (encoder_hidden_states.float() * encoder_attention_mask.float().unsqueeze(-1)).sum(dim=(1, 2))
        ctx = ctx / encoder_attention_mask.float().sum(dim=1).clamp_min(1.0)
        cond = 0.05 * torch.tanh(ctx) + 0.02 * pooled_projections.float().mean(dim=1) + 1e-6 * guidance.float() + bias
        if latent_indices is not None:
            cond = cond + 1e-3 * latent_indices.float().sum()
        cond = cond.view(b, 1, 1, 1, 1)
        x = hidden_states.float()
        return ((0.3 * torch.tanh(x) + cond + 1e-4 * timestep.float().view(b, 1, 1, 1, 1)).to(hidden_states.dtype),)


def _randn(seed, shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def cases():
    """name -> kwargs of sample_hunyuan (without `transformer` and `generator`; the generator is seeded per case)."""
    emb = lambda s, b=1: _randn(s, (b, 12, 8))                                              # noqa: E731
    mask = lambda n, b=1: (torch.arange(12)[None, :] < n).expand(b, 12).clone()               # noqa: E731
    pool = lambda s, b=1: _randn(s, (b, 6))                                                   # noqa: E731
    return {
        "positive_only": dict(width=32, height=48, frames=9, num_inference_steps=5, prompt_embeds=emb(1),
                              prompt_embeds_mask=mask(7), prompt_poolers=pool(2), negative_prompt_embeds=emb(3),
                              negative_prompt_embeds_mask=mask(1), negative_prompt_poolers=pool(4),
                              dtype=torch.float32, real_guidance_scale=1.0, distilled_guidance_scale=10.0),
        "cfg_shift_bf16": dict(width=64, height=32, frames=13, num_inference_steps=4, shift=3.0, prompt_embeds=emb(5),
                               prompt_embeds_mask=mask(12), prompt_poolers=pool(6), negative_prompt_embeds=emb(7),
                               negative_prompt_embeds_mask=mask(3), negative_prompt_poolers=pool(8), dtype=torch.bfloat16,
                               real_guidance_scale=3.0, guidance_rescale=0.5, has_negative_prompt=True),
        "img2img_batch2": dict(width=32, height=32, frames=5, num_inference_steps=6, batch_size=2, strength=0.7,
                               initial_latent=_randn(9, (2, 16, 2, 4, 4)), concat_latent=None, prompt_embeds=emb(10),
                               prompt_embeds_mask=mask(5), prompt_poolers=pool(11), negative_prompt_embeds=emb(12),
                               negative_prompt_embeds_mask=mask(2), negative_prompt_poolers=pool(13),
                               dtype=torch.float32, real_guidance_scale=2.0, has_negative_prompt=True,
                               movement_scale=0.25),
        "extra_kwargs": dict(width=32, height=32, frames=1, num_inference_steps=8, prompt_embeds=emb(14, 2),
                             prompt_embeds_mask=mask(9, 2), prompt_poolers=pool(15, 2), negative_prompt_embeds=emb(16, 2),
                             negative_prompt_embeds_mask=mask(4, 2), negative_prompt_poolers=pool(17, 2),
                             dtype=torch.float32, real_guidance_scale=1.5, has_negative_prompt=True,
                             latent_indices=torch.tensor([[3, 4]]), bias=0.01, negative_kwargs=dict(bias=-0.02)),
    }
