"""Host logic of the sampler mirror (framepack_amd.sampler) against the golden vectors made from the
reference's own uni_pc_fm.py / wrapper.py.  Device-agnostic tensor plumbing: runs on CPU tensors here."""
import os

import numpy as np
import pytest
import torch

from framepack_amd import sampler as S

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "sampler.npz"))


def _toy(x, t, **extra):
    tt = t[(...,) + (None,) * (x.ndim - 1)]
    return x * (1 - tt) * 0.7 + 0.1 * torch.sin(3.0 * x) + 0.05 * tt


@pytest.mark.parametrize("n", [1, 2, 3, 5, 25])
def test_unipc_matches_reference_trajectories(G, n):
    x = torch.randn(2, 4, 6, 5, 8, generator=torch.Generator("cpu").manual_seed(100 + n))
    sig = torch.from_numpy(G[f"unipc_{n}_sigmas"])
    xs = []
    res = S.sample_unipc(_toy, x.clone(), sig, extra_args={"movement_scale": 0.1},
                         callback=lambda d: xs.append(d["x"].clone()), disable=True)
    # host-side fp32 scalars vs the reference's device-side 0-dim tensors: identical ops, a few ulp at most
    torch.testing.assert_close(torch.stack(xs), torch.from_numpy(G[f"unipc_{n}_xs"]), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(res, torch.from_numpy(G[f"unipc_{n}_final"]), rtol=2e-5, atol=2e-6)


def test_k_model_matches_reference(G):
    def stub(hidden_states, timestep, return_dict, bias=0.0, **kw):
        return (torch.tanh(hidden_states.float() * 0.5 + timestep.float().view(-1, 1, 1, 1, 1) * 1e-3 + bias)
                .to(hidden_states.dtype),)
    x = torch.from_numpy(G["kmodel_x"])
    sigma = torch.tensor([0.7, 0.7])
    km = S.fm_wrapper(stub)
    cases = {"pos_only": (1.0, 0.0, False), "cfg": (3.0, 0.0, True), "cfg_rescale": (3.0, 0.7, True)}
    for name, (cs, cr, neg) in cases.items():
        for dn, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            y = km(x, sigma, dtype=dt, cfg_scale=cs, cfg_rescale=cr, concat_latent=None, positive=dict(bias=0.0),
                   negative=dict(bias=-0.3), has_negative_prompt=neg)
            assert np.array_equal(y.numpy(), G[f"kmodel_{name}_{dn}"]), (name, dn)


def test_flux_schedule():
    assert abs(S.calculate_flux_mu(14400) - 1.9459101) < 1e-6
    assert abs(S.calculate_flux_mu(768) - 0.586667) < 1e-5
    s = S.get_flux_sigmas_from_mu(25, S.calculate_flux_mu(14400))
    assert s.shape == (26,) and s[0] == 1.0 and s[-1] == 0.0


def test_sample_hunyuan_plumbing_with_stub_transformer():
    """sample_hunyuan draws the noise exactly as the reference (CPU generator), builds kwargs and runs
    `num_inference_steps` model calls."""
    calls = []

    class Stub:
        device = torch.device("cpu")

        def __call__(self, hidden_states, timestep, return_dict, **kw):
            calls.append((float(timestep[0]), sorted(kw)))
            return (hidden_states * 0.1,)

    g = torch.Generator("cpu").manual_seed(31337)
    out = S.sample_hunyuan(Stub(), width=64, height=64, frames=9, num_inference_steps=4, generator=g,
                           distilled_guidance_scale=10.0, prompt_embeds=torch.zeros(1, 8, 16),
                           prompt_embeds_mask=torch.ones(1, 8, dtype=torch.bool), prompt_poolers=torch.zeros(1, 4),
                           dtype=torch.bfloat16, device="cpu", latent_indices=torch.arange(3)[None])
    assert out.shape == (1, 16, 3, 8, 8) and len(calls) == 4
    assert calls[0][0] == 1000.0 and "latent_indices" in calls[0][1] and "guidance" in calls[0][1]
    want = torch.randn((1, 16, 3, 8, 8), generator=torch.Generator("cpu").manual_seed(31337))
    assert torch.isfinite(out).all() and want.shape == out.shape


def test_sample_hunyuan_matches_the_reference_pipeline(golden_dir):
    """tests/golden/pipeline.npz: the REFERENCE's sample_hunyuan (k_diffusion_hunyuan.py:27-124) run end to end on the
    seeded cases and the stub transformer of tests/pipeline_cases.py - noise draw, flux-shift schedule, image-to-image
    blend, batch repetition, CFG with rescale, extra / negative kwargs, the UniPC steps and their callback.  The first
    three steps' denoised estimates (noise, schedule, argument packing, k-model, first-order updates) must be the same
    BITS; from the third step on the mirror sums the two UniPC history terms as plain weighted sums where the reference
    calls tensordot (sampler.py), which moves the last bit: same tolerance as the trajectory test above."""
    import pipeline_cases as PC
    P = np.load(os.path.join(golden_dir, "pipeline.npz"))
    for i, (name, kw) in enumerate(PC.cases().items()):
        steps = []
        res = S.sample_hunyuan(transformer=PC.StubTransformer(), generator=torch.Generator().manual_seed(40 + i),
                               callback=lambda d: steps.append(d["denoised"].float().clone().numpy()), **kw)
        got, want = np.stack(steps), P[f"{name}/denoised"]
        assert got.shape == want.shape and np.array_equal(got[:3], want[:3]), name
        torch.testing.assert_close(torch.from_numpy(got), torch.from_numpy(want), rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(res.float(), torch.from_numpy(P[f"{name}/result"]), rtol=2e-5, atol=2e-6)
    for n in (1, 1024, 14400, 100000):
        assert S.calculate_flux_mu(n, exp_max=7.0) == float(P[f"mu/{n}"][0])
    assert np.array_equal(S.get_flux_sigmas_from_mu(25, 1.2).numpy(), P["sigmas/25_mu1.2"])
