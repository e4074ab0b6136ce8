"""Seeded state dicts and LoRA files for the LoRA-merge goldens (shared by tests/golden/make_golden.py, which feeds
them to the reference's utils/lora_utils.py, and tests/test_weights_cpu.py, which feeds them to framepack_amd.weights)."""
import torch

CASES = ["musubi", "diffusion_pipe", "hunyuan", "unknown"]


def _r(g, *shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def sample(w: torch.Tensor) -> torch.Tensor:
    """What the golden stores of a merged weight: a strided sub-grid, as fp64."""
    w2 = w.reshape(w.shape[0], -1)
    return w2[:: max(1, w2.shape[0] // 16), :: max(1, w2.shape[1] // 24)].double()


def build(name: str):
    g = torch.Generator("cpu").manual_seed({"musubi": 1, "diffusion_pipe": 2, "hunyuan": 3, "unknown": 4}[name])
    d = 64
    if name == "musubi":
        sd = {"transformer_blocks.0.attn.to_q.weight": _r(g, d, d, dtype=torch.bfloat16, scale=0.05),
              "transformer_blocks.0.attn.to_q.bias": _r(g, d, dtype=torch.bfloat16),
              "transformer_blocks.0.ff.net.0.proj.weight": _r(g, 4 * d, d, dtype=torch.bfloat16, scale=0.05),
              "single_transformer_blocks.1.proj_out.weight": _r(g, d, 5 * d, scale=0.05),
              "x_embedder.proj.weight": _r(g, d, 16, 1, 2, 2, scale=0.05)}
        p = "lora_unet_"
        lora = {p + "transformer_blocks_0_attn_to_q.lora_down.weight": _r(g, 4, d, scale=0.3),
                p + "transformer_blocks_0_attn_to_q.lora_up.weight": _r(g, d, 4, scale=0.3),
                p + "transformer_blocks_0_attn_to_q.alpha": torch.tensor(2.0),
                p + "transformer_blocks_0_ff_net_0_proj.lora_down.weight": _r(g, 8, d, dtype=torch.bfloat16, scale=0.3),
                p + "transformer_blocks_0_ff_net_0_proj.lora_up.weight": _r(g, 4 * d, 8, dtype=torch.bfloat16, scale=0.3),
                # no alpha: alpha = rank
                p + "single_transformer_blocks_1_proj_out.lora_down.weight": _r(g, 2, 5 * d, scale=0.3),
                p + "single_transformer_blocks_1_proj_out.lora_up.weight": _r(g, d, 2, scale=0.3),
                p + "single_transformer_blocks_1_proj_out.alpha": torch.tensor(8.0),
                # a module the state dict does not have: skipped with a message
                p + "transformer_blocks_7_attn_to_v.lora_down.weight": _r(g, 2, d),
                p + "transformer_blocks_7_attn_to_v.lora_up.weight": _r(g, d, 2)}
        return sd, lora, 0.8
    if name == "diffusion_pipe":
        sd = {"transformer_blocks.0.attn.to_k.weight": _r(g, d, d, dtype=torch.bfloat16, scale=0.05),
              "transformer_blocks.0.attn.add_q_proj.weight": _r(g, d, d, scale=0.05),
              "norm_out.linear.weight": _r(g, 2 * d, d, scale=0.05)}
        lora = {"text_encoder.layers.0.q_proj.lora_A.weight": _r(g, 4, 32),              # ignored (message)
                "text_encoder.layers.0.q_proj.lora_B.weight": _r(g, 32, 4),
                "diffusion_model.transformer_blocks.0.attn.to_k.lora_A.weight": _r(g, 4, d, scale=0.3),
                "diffusion_model.transformer_blocks.0.attn.to_k.lora_B.weight": _r(g, d, 4, scale=0.3),
                "diffusion_model.transformer_blocks.0.attn.add_q_proj.lora_A.weight": _r(g, 16, d, scale=0.3),
                "diffusion_model.transformer_blocks.0.attn.add_q_proj.lora_B.weight": _r(g, d, 16, scale=0.3),
                "diffusion_model.norm_out.linear.lora_A.weight": _r(g, 2, d, scale=0.3),
                "diffusion_model.norm_out.linear.lora_B.weight": _r(g, 2 * d, 2, scale=0.3)}
        return sd, lora, 1.0
    if name == "hunyuan":
        # the reference asserts the real widths (3072 / 9216 / 21504): one double and one single block at full width
        D, M = 3072, 12288
        bf = torch.bfloat16
        sd = {}
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            sd[f"transformer_blocks.0.attn.{n}.weight"] = _r(g, D, D, dtype=bf, scale=0.02)
        sd["transformer_blocks.0.ff.net.0.proj.weight"] = _r(g, M, D, dtype=bf, scale=0.02)
        sd["transformer_blocks.0.norm1.linear.weight"] = _r(g, 6 * D, D, dtype=bf, scale=0.02)
        for n in ("attn.to_q", "attn.to_k", "attn.to_v"):
            sd[f"single_transformer_blocks.0.{n}.weight"] = _r(g, D, D, dtype=bf, scale=0.02)
        sd["single_transformer_blocks.0.proj_mlp.weight"] = _r(g, M, D, dtype=bf, scale=0.02)
        sd["single_transformer_blocks.0.proj_out.weight"] = _r(g, D, D + M, dtype=bf, scale=0.02)
        p = "lora_unet_"
        lora = {}

        def add(mod, n_in, n_out, r, alpha=None):
            lora[p + mod + ".lora_down.weight"] = _r(g, r, n_in, dtype=bf, scale=0.1)
            lora[p + mod + ".lora_up.weight"] = _r(g, n_out, r, dtype=bf, scale=0.1)
            if alpha is not None:
                lora[p + mod + ".alpha"] = torch.tensor(float(alpha))
        add("double_blocks_0_img_attn_qkv", D, 3 * D, 4, alpha=4)
        add("double_blocks_0_txt_attn_qkv", D, 3 * D, 2, alpha=1)
        add("double_blocks_0_img_attn_proj", D, D, 2)
        add("double_blocks_0_img_mlp_fc1", D, M, 2, alpha=2)
        add("double_blocks_0_img_mod_linear", D, 6 * D, 2, alpha=2)
        add("single_blocks_0_linear1", D, 3 * D + M, 4, alpha=2)
        add("single_blocks_0_linear2", D + M, D, 2, alpha=2)
        return sd, lora, 0.5
    if name == "unknown":
        sd = {"proj_out.weight": _r(g, 8, d)}
        lora = {"foo.bar.weight": _r(g, 2, 2)}
        return sd, lora, 1.0
    raise KeyError(name)
