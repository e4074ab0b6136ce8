"""framepack_amd.weights (LoRA merge) against goldens made by the reference's own
utils/lora_utils.py on the same seeded inputs (tests/golden/lora.npz <- tests/golden/make_golden.py)."""
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
