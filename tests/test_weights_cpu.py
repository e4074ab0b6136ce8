"""framepack_amd.weights (LoRA merge and FP8 checkpoint optimisation) against goldens made by
the reference's own utils/lora_utils.py and utils/fp8_optimization_utils.py on the same seeded inputs
(tests/golden/lora.npz, fp8.npz <- tests/golden/make_golden.py)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

import lora_cases
from framepack_amd import weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lora.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name", lora_cases.CASES)
def test_merge_matches_reference(golden, name, tmp_path):
    sd, lora, mult = lora_cases.build(name)
    before = {k: v.clone() for k, v in sd.items()}
    path = str(tmp_path / (name + ".safetensors"))
    save_file({k: v.contiguous() for k, v in lora.items()}, path)
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        merged = weights.merge_lora_to_state_dict(sd, path, mult, torch.device("cpu"))
    assert set(merged) == set(before)
    n_changed = 0
    for k, v in merged.items():
        changed = bool(golden[f"{name}/{k}/changed"][0])
        assert str(v.dtype) == str(golden[f"{name}/{k}/dtype"][0]), k
        if not changed:
            assert v.dtype == before[k].dtype and torch.equal(v, before[k]), k
            continue
        n_changed += 1
        # same operations in the same order and dtypes: bit-exact
        assert np.array_equal(lora_cases.sample(v).numpy(), golden[f"{name}/{k}/sample"]), k
        s = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        assert np.array_equal(s, golden[f"{name}/{k}/sum"]), k
    expect = {"musubi": 3, "diffusion_pipe": 3, "hunyuan": 14, "unknown": 0}[name]
    assert n_changed == expect
    text = log.getvalue()
    if name == "musubi":
        assert "No module found for LoRA weight" in text
    if name == "diffusion_pipe":
        assert "unexpected key: text_encoder" in text
    if name == "hunyuan":
        assert "HunyuanVideo LoRA detected" in text
    if name == "unknown":
        assert "LoRA file format not recognized" in text


def test_format_detection():
    assert weights.detect_lora_format(["lora_unet_a.lora_down.weight"]) == "musubi"
    assert weights.detect_lora_format(["transformer.a.lora_A.weight", "transformer.a.lora_B.weight"]) == "diffusion_pipe"
    assert weights.detect_lora_format(["text_encoder.x.lora_A.weight"]) is None          # no transformer prefix
    assert weights.detect_lora_format(["diffusion_model.a.lora_down.weight"]) is None     # prefix but no lora_A
    assert weights.detect_lora_format([]) is None


def test_hunyuan_conversion_splits_by_the_down_width():
    """Reduced width (the reference hard-codes 3072 / 21504): the cut points follow the module's input width."""
    d, m, r = 16, 64, 2
    g = torch.Generator().manual_seed(0)
    up = torch.randn(3 * d + m, r, generator=g)
    lora = {"lora_unet_single_blocks_3_linear1.lora_down.weight": torch.randn(r, d, generator=g),
            "lora_unet_single_blocks_3_linear1.lora_up.weight": up,
            "lora_unet_single_blocks_3_linear1.alpha": torch.tensor(1.0)}
    out = weights.convert_hunyuan_to_framepack(lora)
    base = "lora_unet_single_transformer_blocks_3_"
    assert torch.equal(out[base + "attn_to_q.lora_up.weight"], up[:d])
    assert torch.equal(out[base + "attn_to_k.lora_up.weight"], up[d:2 * d])
    assert torch.equal(out[base + "attn_to_v.lora_up.weight"], up[2 * d:3 * d])
    assert torch.equal(out[base + "proj_mlp.lora_up.weight"], up[3 * d:])
    for n in ("attn_to_q", "attn_to_k", "attn_to_v", "proj_mlp"):
        assert out[base + n + ".alpha"].item() == 1.0
        assert out[base + n + ".lora_down.weight"].shape == (r, d)


def test_merged_state_dict_loads_into_the_model():
    from framepack_amd import synthetic
    from framepack_amd.model import HunyuanVideoTransformer3DModelPacked
    m = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd0 = sd["transformer_blocks.0.attn.to_q.weight"].clone()
    d = synthetic.CONFIG_TINY["num_attention_heads"] * synthetic.CONFIG_TINY["attention_head_dim"]
    lora = {"transformer.transformer_blocks.0.attn.to_q.lora_A.weight": torch.randn(4, d) * 0.1,
            "transformer.transformer_blocks.0.attn.to_q.lora_B.weight": torch.randn(d, 4) * 0.1}
    with contextlib.redirect_stdout(io.StringIO()):
        merged = weights.merge_lora_to_state_dict(sd, lora, 1.0, torch.device("cpu"))
    m.load_state_dict(merged, strict=True)
    delta = lora["transformer.transformer_blocks.0.attn.to_q.lora_B.weight"] @ lora["transformer.transformer_blocks.0.attn.to_q.lora_A.weight"]
    got = m.state_dict()["transformer_blocks.0.attn.to_q.weight"]
    assert torch.allclose(got, sd0 + delta, atol=1e-6)


# ------------------------------------------------------------------------------------------------ FP8 checkpoints
def _fp8_state_dict(dt):
    G = np.load(os.path.join(os.path.dirname(GOLDEN), "fp8.npz"))
    w = torch.from_numpy(G["w"])
    sd = {"transformer_blocks.0.attn.to_q.weight": w.to(dt), "transformer_blocks.0.attn.to_q.bias": torch.zeros(48, dtype=dt),
          "transformer_blocks.0.attn.norm_q.weight": torch.ones(8, dtype=dt),
          "transformer_blocks.0.norm1.linear.weight": w.to(dt).clone(), "proj_out.weight": w.to(dt).clone()}
    return G, sd


@pytest.mark.parametrize("dt_name,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_fp8_optimisation_matches_reference_bytes_and_scale(dt_name, dt):
    G, sd = _fp8_state_dict(dt)
    with contextlib.redirect_stdout(io.StringIO()):
        q = weights.optimize_state_dict_with_fp8(sd, None, ["transformer_blocks", "single_transformer_blocks"], ["norm"])
    assert q is sd                                                      # in place, as the reference
    assert sorted(q.keys()) == [str(k) for k in G[f"{dt_name}_keys"]]
    k = "transformer_blocks.0.attn.to_q"
    assert q[k + ".weight"].dtype == torch.float8_e4m3fn
    assert np.array_equal(q[k + ".weight"].view(torch.uint8).numpy(), G[f"{dt_name}_w8_bytes"])
    assert q[k + ".scale_weight"].dtype == dt and q[k + ".scale_weight"].shape == (1,)
    assert np.array_equal(q[k + ".scale_weight"].float().numpy(), G[f"{dt_name}_scale"])
    assert q["transformer_blocks.0.norm1.linear.weight"].dtype == dt    # excluded by "norm"
    assert q["proj_out.weight"].dtype == dt                             # not a targeted block


def test_fp8_maxval_and_argument_checks():
    G = np.load(os.path.join(os.path.dirname(GOLDEN), "fp8.npz"))
    assert weights.calculate_fp8_maxval(4, 3) == float(G["maxval_e4m3"][0]) == 448.0
    assert weights.calculate_fp8_maxval(5, 2) == float(G["maxval_e5m2"][0])
    with pytest.raises(AssertionError):
        weights.calculate_fp8_maxval(4, 4)
    with pytest.raises(ValueError):
        weights.optimize_state_dict_with_fp8({}, None, exp_bits=3, mantissa_bits=4)
    t = torch.tensor([0.0, 1e-9, -500.0, 0.3])
    q, s = weights.quantize_tensor_to_fp8(t, 1.0)
    assert s == 1.0 and q[0] == 0 and q[2] == -448.0 and q[1] == 0      # clamp, and the subnormal step rounds 1e-9 to 0


def test_fp8_patch_registers_scale_buffers_for_assign_load():
    from framepack_amd.model import HunyuanVideoTransformer3DModelPacked
    from framepack_amd import synthetic
    m = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY).to(torch.bfloat16)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with contextlib.redirect_stdout(io.StringIO()):
        q = weights.optimize_state_dict_with_fp8(sd, None, ["transformer_blocks", "single_transformer_blocks"], ["norm"])
        weights.apply_fp8_monkey_patch(m, q)
    n_scales = sum(k.endswith(".scale_weight") for k in q)
    assert n_scales > 0 and sum(hasattr(mod, "scale_weight") for mod in m.modules()) == n_scales
    m.load_state_dict(q, strict=True, assign=True)
    lin = m.transformer_blocks[0].attn.to_q
    # (torch's loader takes a [1] tensor into a 0-dim buffer: the scale arrives as a scalar, as in the reference)
    assert lin.weight.dtype == torch.float8_e4m3fn and lin.scale_weight.numel() == 1
    assert lin.scale_weight.item() == q["transformer_blocks.0.attn.to_q.scale_weight"].item()
    assert m.transformer_blocks[0].norm1.linear.weight.dtype == torch.bfloat16


def test_load_transformer_sequence_lora_then_fp8(tmp_path):
    """ref framepack/models.py:233-275: directory load, LoRA merged into the state dict, FP8 on top, assign-load."""
    import json
    from framepack_amd.model import HunyuanVideoTransformer3DModelPacked
    from framepack_amd import synthetic
    torch.manual_seed(0)
    base = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY).to(torch.bfloat16)
    mdir = tmp_path / "model"
    mdir.mkdir()
    (mdir / "config.json").write_text(json.dumps(dict(synthetic.CONFIG_TINY, _class_name="HunyuanVideoTransformer3DModelPacked")))
    save_file({k: v.contiguous() for k, v in base.state_dict().items()}, str(mdir / "diffusion_pytorch_model.safetensors"))
    d = base.transformer_blocks[0].attn.to_q.weight.shape[1]
    g = torch.Generator().manual_seed(1)
    lora = {"lora_unet_transformer_blocks_0_attn_to_q.lora_down.weight": torch.randn(4, d, generator=g) * 0.1,
            "lora_unet_transformer_blocks_0_attn_to_q.lora_up.weight": torch.randn(d, 4, generator=g) * 0.1,
            "lora_unet_transformer_blocks_0_attn_to_q.alpha": torch.tensor(2.0)}
    lpath = str(tmp_path / "l.safetensors")
    save_file(lora, lpath)
    with contextlib.redirect_stdout(io.StringIO()) as log:
        plain = weights.load_transformer(str(mdir))
        missing = weights.load_transformer(str(mdir), lora_file=str(tmp_path / "absent.safetensors"))
        both = weights.load_transformer(str(mdir), lora_file=lpath, lora_multiplier=0.8, fp8_optimization=True)
    assert "not found, skipping" in log.getvalue()
    w0 = base.transformer_blocks[0].attn.to_q.weight
    assert plain.high_quality_fp32_output_for_inference and torch.equal(plain.transformer_blocks[0].attn.to_q.weight, w0)
    assert torch.equal(missing.transformer_blocks[0].attn.to_q.weight, w0)
    lin = both.transformer_blocks[0].attn.to_q
    assert lin.weight.dtype == torch.float8_e4m3fn
    # merged in the reference's arithmetic (fp32 LoRA on a bf16 weight gives fp32), then one scale per tensor
    merged = w0 + 0.8 * (lora["lora_unet_transformer_blocks_0_attn_to_q.lora_up.weight"]
                         @ lora["lora_unet_transformer_blocks_0_attn_to_q.lora_down.weight"]) * (2.0 / 4)
    scale = merged.abs().max() / 448.0
    assert lin.scale_weight.dtype == merged.dtype and torch.equal(lin.scale_weight.reshape(()), scale)
    deq = lin.weight.to(merged.dtype) * lin.scale_weight
    assert (deq - merged).abs().max() <= scale * 16 + 1e-6            # half of the coarsest E4M3 step (32)
    other = both.transformer_blocks[0].attn.to_k                       # no LoRA: quantised base weight
    assert other.weight.dtype == torch.float8_e4m3fn and other.scale_weight.dtype == torch.bfloat16
    assert both.x_embedder.proj.weight.dtype == torch.bfloat16          # outside the targeted blocks
