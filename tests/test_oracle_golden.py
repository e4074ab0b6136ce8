"""The oracle against the golden vectors made from the reference's own modules
(tests/golden/make_golden.py).  CPU only.  Bit-exact where the oracle restates torch ops in the
same order (it does everywhere), so tolerances are 0 unless stated."""
import os

import numpy as np
import pytest
import torch

from framepack_amd import synthetic
from oracle import dit_oracle as O
from oracle import fp8_oracle as F8
from oracle import sampler_oracle as SO

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def rope_g(golden_dir):
    return np.load(os.path.join(golden_dir, "rope.npz"))


@pytest.fixture(scope="module")
def tiny(golden_dir):
    cfg = O.DiTConfig(**synthetic.CONFIG_TINY)
    return cfg, O.synth_params(cfg, seed=0), np.load(os.path.join(golden_dir, "dit_tiny.npz"))


@pytest.mark.parametrize("name", ["noisy", "clean", "x2", "x4", "x2_odd", "x4_odd", "x4_pad"])
def test_rope_tables_bit_exact(rope_g, name):
    cfg = O.DiTConfig()
    idx = torch.from_numpy(rope_g[name + "_idx"])
    h, w = (int(v) for v in rope_g[name + "_hw"])
    t = O.rope_table(cfg, idx, h, w)
    if name.startswith("x2"):
        t = O.center_down_sample_3d(O.pad_for_3d_conv(t, (2, 2, 2)), (2, 2, 2))
    if name.startswith("x4"):
        t = O.center_down_sample_3d(O.pad_for_3d_conv(t, (4, 4, 4)), (4, 4, 4))
    got = t.flatten(2).transpose(1, 2).numpy()
    assert got.shape == rope_g[name].shape
    assert np.array_equal(got, rope_g[name])


def test_apply_rope_and_pad(rope_g):
    g = torch.Generator("cpu").manual_seed(int(rope_g["apply_in_seed"][0]))
    x = torch.randn(1, 48 * 9, 3, 128, generator=g).to(torch.bfloat16)
    y = O.apply_rope(x, torch.from_numpy(rope_g["noisy"]))
    assert np.array_equal(y.float().numpy(), rope_g["apply_out"])
    r = torch.arange(2 * 3 * 5 * 7, dtype=torch.float32).view(1, 2, 3, 5, 7)
    assert np.array_equal(O.pad_for_3d_conv(r, (2, 4, 8)).numpy(), rope_g["pad_248"])


def _sub(t, tok=97, ch=5):
    return t[0, ::tok, ::ch].float().numpy()


def test_dit_tiny_forward_and_block_taps(tiny):
    cfg, P, G = tiny
    kw = synthetic.make_inputs(seed=31337, frames_latent=3, height=256, width=256, n_text=37,
                               dtype=torch.float32, sigma=0.9375)
    taps = {}
    y = O.dit_forward(P, cfg, taps=taps, **kw)
    assert np.array_equal(y.numpy(), G["A_out"])
    assert list(taps["packed_hidden"].shape) == list(G["A_packed_shape"]) == [1, 1408, 256]
    assert np.array_equal(_sub(taps["packed_hidden"]), G["A_packed_h"])
    assert np.array_equal(_sub(taps["rope"], 31, 3), G["A_packed_rope"])
    for kind, n in (("double", cfg.num_layers), ("single", cfg.num_single_layers)):
        for i in range(n):
            h, e = taps[f"{kind}.{i}"]
            assert np.array_equal(_sub(h), G[f"A_{kind}.{i}_h"])
            assert np.array_equal(_sub(e, tok=29), G[f"A_{kind}.{i}_e"])
            np.testing.assert_allclose([h.norm().item(), e.norm().item()], G[f"A_{kind}.{i}_norms"], rtol=1e-6)


def test_dit_tiny_odd_bucket_padding_fp32_head(tiny):
    cfg, P, G = tiny
    kw = synthetic.make_inputs(seed=5, frames_latent=2, height=160, width=304, n_text=5, dtype=torch.float32,
                               sigma=0.31, latent_padding=1)
    y = O.dit_forward(P, cfg, fp32_output=True, **kw)
    assert np.array_equal(y.numpy(), G["B_out"])


def test_dit_tiny_one_frame_mode(tiny):
    cfg, P, G = tiny
    kw = synthetic.make_inputs(seed=6, frames_latent=1, height=128, width=128, n_text=256, dtype=torch.float32,
                               sigma=0.77, with_2x=False, with_4x=False)
    assert np.array_equal(O.dit_forward(P, cfg, **kw).numpy(), G["C_out"])


def test_dit_tiny_teacache_schedule(tiny):
    cfg, P, G = tiny
    kw = synthetic.make_inputs(seed=31337, frames_latent=3, height=256, width=256, n_text=37,
                               dtype=torch.float32, sigma=0.9375)
    tc = O.TeaCacheState(enable=True, num_steps=6, rel_l1_thresh=0.15)
    for j, s in enumerate(G["D_sigmas"]):
        kw["timestep"] = torch.tensor([float(s) * 1000.0])
        y = O.dit_forward(P, cfg, teacache=tc, **kw)
        assert np.array_equal(y[0, :, :, ::4, ::4].numpy(), G["D_outs"][j])
        assert [tc.hits, tc.misses, tc.queries] == list(G["D_counts"][j])
    assert tc.hits > 0 and tc.misses > 0


def test_dit_tiny_bf16_rounding_points(tiny):
    cfg, P, G = tiny
    Pb = {k: v.to(torch.bfloat16) for k, v in P.items()}
    kw = synthetic.make_inputs(seed=31337, frames_latent=3, height=256, width=256, n_text=37,
                               dtype=torch.bfloat16, sigma=0.9375)
    y = O.dit_forward(Pb, cfg, **kw)
    assert y.dtype == torch.bfloat16
    assert np.array_equal(y.float().numpy(), G["E_out_bf16"])


def test_cu_seqlens_two_segments_per_sample():
    mask = torch.zeros(2, 8, dtype=torch.bool)
    mask[0, :3] = True
    mask[1, :8] = True
    cu = O.get_cu_seqlens(mask, 10)
    assert cu.dtype == torch.int32
    assert cu.tolist() == [0, 13, 18, 18 + 18, 36]


# ------------------------------------------------------------------ sampler
@pytest.fixture(scope="module")
def samp(golden_dir):
    return np.load(os.path.join(golden_dir, "sampler.npz"))


def _toy(x, t):
    tt = t[(...,) + (None,) * (x.ndim - 1)]
    return x * (1 - tt) * 0.7 + 0.1 * torch.sin(3.0 * x) + 0.05 * tt


@pytest.mark.parametrize("n", [1, 2, 3, 5, 25])
def test_unipc_trajectories(samp, n):
    x = torch.randn(2, 4, 6, 5, 8, generator=torch.Generator("cpu").manual_seed(100 + n))
    sig = torch.from_numpy(samp[f"unipc_{n}_sigmas"])
    xs = []
    res = SO.unipc_sample(_toy, x.clone(), sig, 0.1, callback=lambda d: xs.append(d["x"].clone().numpy()))
    assert np.array_equal(np.stack(xs), samp[f"unipc_{n}_xs"])
    assert np.array_equal(res.numpy(), samp[f"unipc_{n}_final"])


def test_k_model_cfg_paths(samp):
    def stub(hidden_states, timestep, return_dict, bias=0.0, **kw):
        return (torch.tanh(hidden_states.float() * 0.5 + timestep.float().view(-1, 1, 1, 1, 1) * 1e-3 + bias)
                .to(hidden_states.dtype),)
    x = torch.from_numpy(samp["kmodel_x"])
    sigma = torch.tensor([0.7, 0.7])
    cases = {"pos_only": (1.0, 0.0, False), "cfg": (3.0, 0.0, True), "cfg_rescale": (3.0, 0.7, True)}
    for name, (cs, cr, neg) in cases.items():
        for dn, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            y = SO.k_model(stub, x, sigma, dtype=dt, cfg_scale=cs, cfg_rescale=cr, positive=dict(bias=0.0),
                           negative=dict(bias=-0.3), has_negative_prompt=neg)
            assert np.array_equal(y.numpy(), samp[f"kmodel_{name}_{dn}"]), (name, dn)


def test_flux_sigma_schedule_known_values():
    # L=14400 -> mu clamps to ln 7; L=768 -> 0.586667
    assert abs(SO.calculate_flux_mu(14400) - 1.9459101) < 1e-6
    assert abs(SO.calculate_flux_mu(768) - 0.586667) < 1e-5
    s = SO.flux_sigmas(25, SO.calculate_flux_mu(14400))
    assert s.shape == (26,) and s[0] == 1.0 and s[-1] == 0.0 and bool((s[:-1] > s[1:]).all())


# ------------------------------------------------------------------ fp8
def test_fp8_quantisation_bytes_scales_and_linear(golden_dir):
    G = np.load(os.path.join(golden_dir, "fp8.npz"))
    assert F8.fp8_maxval(4, 3) == float(G["maxval_e4m3"][0]) == 448.0
    assert F8.fp8_maxval(5, 2) == float(G["maxval_e5m2"][0])
    w = torch.from_numpy(G["w"])
    for dn, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        sd = {"transformer_blocks.0.attn.to_q.weight": w.to(dt),
              "transformer_blocks.0.attn.to_q.bias": torch.zeros(48, dtype=dt),
              "transformer_blocks.0.attn.norm_q.weight": torch.ones(8, dtype=dt),
              "transformer_blocks.0.norm1.linear.weight": w.to(dt).clone(),
              "proj_out.weight": w.to(dt).clone()}
        q = F8.quantize_state_dict(sd)
        assert sorted(q.keys()) == list(G[f"{dn}_keys"])
        k = "transformer_blocks.0.attn.to_q"
        assert q[k + ".weight"].dtype == torch.float8_e4m3fn
        assert np.array_equal(q[k + ".weight"].view(torch.uint8).numpy(), G[f"{dn}_w8_bytes"])
        assert q[k + ".scale_weight"].dtype == dt
        assert np.array_equal(q[k + ".scale_weight"].float().numpy(), G[f"{dn}_scale"])
        x = torch.randn(1, 5, 64, generator=torch.Generator("cpu").manual_seed(4)).to(dt)
        y = F8.fp8_linear(x, q[k + ".weight"], q[k + ".scale_weight"], sd[k + ".bias"])
        assert np.array_equal(y.float().numpy(), G[f"{dn}_linear_out"])
