"""Weight formats around the DiT: LoRA merge into a FramePack state dict.

Host-side preprocessing with the reference's names, arguments and results:
    merge_lora_to_state_dict(state_dict, lora_file, multiplier, device)       ref utils/lora_utils.py:7-39
accepting the three layouts the reference accepts -
  * Musubi Tuner            `lora_unet_<module path with _>.lora_down.weight / .lora_up.weight / .alpha`      (ref :76-157)
  * diffusion-pipe          `diffusion_model.<module path>.lora_A.weight / .lora_B.weight` (alpha = rank)     (ref :42-73)
  * HunyuanVideo modules    `double_blocks` / `single_blocks` names with fused QKV / QKV+MLP up-projections,
                            renamed and split to FramePack's per-projection modules                            (ref :160-234)
The arithmetic is W <- W + multiplier * (up @ down) * (alpha / rank), evaluated with the reference's operand order
and dtypes (a bf16 weight merged with an fp32 LoRA comes back fp32, as in the reference).  The fused-projection
widths are taken from the tensors (the reference hard-codes 3072 / 21504), so reduced-width test models work too.

The FP8 checkpoint layout (`.scale_weight`) needs no converter: HunyuanVideoTransformer3DModelPacked consumes the
reference's own format (model.py `_dense_weight` / `_fuse_fp8`).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Union

import torch

Tensor = torch.Tensor
_PREFIX = "lora_unet_"


def load_lora(lora_file: Union[str, Dict[str, Tensor]]) -> Dict[str, Tensor]:
    if isinstance(lora_file, dict):
        return lora_file
    from safetensors.torch import load_file
    return load_file(lora_file)


def detect_lora_format(keys) -> Optional[str]:
    """'musubi' | 'diffusion_pipe' | None, decided as ref :16-39 decides (first key, then a scan)."""
    keys = list(keys)
    if not keys:
        return None
    if keys[0].startswith(_PREFIX):
        return "musubi"
    has_a = False
    prefix = None
    for key in keys:
        has_a = has_a or "lora_A" in key
        if prefix is None and key.split(".")[0] in ("diffusion_model", "transformer"):
            prefix = key.split(".")[0]
        if has_a and prefix is not None:
            return "diffusion_pipe"
    return None


def merge_lora_to_state_dict(state_dict: Dict[str, Tensor], lora_file, multiplier: float,
                             device: torch.device) -> Dict[str, Tensor]:
    lora_sd = load_lora(lora_file)
    fmt = detect_lora_format(lora_sd.keys())
    if fmt == "musubi":
        return merge_musubi_tuner(lora_sd, state_dict, multiplier, device)
    if fmt == "diffusion_pipe":
        return merge_diffusion_pipe_or_something(lora_sd, state_dict, _PREFIX, multiplier, device)
    name = lora_file if isinstance(lora_file, str) else "<dict>"
    print(f"LoRA file format not recognized: {os.path.basename(name)}")
    return state_dict


def merge_diffusion_pipe_or_something(lora_sd: Dict[str, Tensor], state_dict: Dict[str, Tensor], prefix: str,
                                      multiplier: float, device: torch.device) -> Dict[str, Tensor]:
    """`<diffusion_model|transformer>.a.b.lora_A.weight` -> `<prefix>a_b.lora_down.weight`, alpha = rank (ref :42-73)."""
    out: Dict[str, Tensor] = {}
    rank: Dict[str, int] = {}
    for key, w in lora_sd.items():
        head, body = key.split(".", 1)
        if head not in ("diffusion_model", "transformer"):
            print(f"unexpected key: {key} in diffusers format")
            continue
        new_key = (prefix + body).replace(".", "_").replace("_lora_A_", ".lora_down.").replace("_lora_B_", ".lora_up.")
        out[new_key] = w
        module = new_key.split(".")[0]
        if "lora_down" in new_key and module not in rank:
            rank[module] = w.shape[0]
    for module, r in rank.items():
        out[f"{module}.alpha"] = torch.tensor(r)
    return merge_musubi_tuner(out, state_dict, multiplier, device)


def merge_musubi_tuner(lora_sd: Dict[str, Tensor], state_dict: Dict[str, Tensor], multiplier: float,
                       device: torch.device) -> Dict[str, Tensor]:
    """ref :76-157."""
    if any(("double_blocks" in k) or ("single_blocks" in k) for k in lora_sd):
        print("HunyuanVideo LoRA detected, converting to FramePack format")
        lora_sd = convert_hunyuan_to_framepack(lora_sd)
    print(f"Merging LoRA weights into state dict. multiplier: {multiplier}")

    target: Dict[str, str] = {}
    for key in state_dict:
        if key.endswith(".weight"):
            target.setdefault(_PREFIX + key[: -len(".weight")].replace(".", "_"), key)

    for key in [k for k in lora_sd if "lora_down" in k]:
        module = ".".join(key.split(".")[:-2])
        if module not in target:
            print(f"No module found for LoRA weight: {key}")
            continue
        down = lora_sd[key]
        up = lora_sd[key.replace("lora_down", "lora_up")]
        r = down.size()[0]
        scale = lora_sd.get(key[: key.index("lora_down")] + "alpha", r) / r
        name = target[module]
        w = state_dict[name]
        home = w.device
        if home != device:
            w = w.to(device)
        down, up = down.to(device), up.to(device)
        if w.dim() == 2:
            if up.dim() == 4:                          # a conv-shaped LoRA on a linear layer
                up, down = up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)
            w = w + multiplier * (up @ down) * scale
        elif tuple(down.size()[2:4]) == (1, 1):        # 1x1 conv
            w = w + multiplier * (up.squeeze(3).squeeze(2) @ down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3) * scale
        else:                                          # 3x3 conv
            w = w + multiplier * torch.nn.functional.conv2d(down.permute(1, 0, 2, 3), up).permute(1, 0, 2, 3) * scale
        state_dict[name] = w.to(home)
    return state_dict


# HunyuanVideo module name -> FramePack module name (ref :166-182); "QKV" / "QKVM" mark fused projections
_DOUBLE = [("double_blocks", "transformer_blocks"), ("img_mod_linear", "norm1_linear"), ("img_attn_qkv", "attn_to_QKV"),
           ("img_attn_proj", "attn_to_out_0"), ("img_mlp_fc1", "ff_net_0_proj"), ("img_mlp_fc2", "ff_net_2"),
           ("txt_mod_linear", "norm1_context_linear"), ("txt_attn_qkv", "attn_add_QKV_proj"),
           ("txt_attn_proj", "attn_to_add_out"), ("txt_mlp_fc1", "ff_context_net_0_proj"),
           ("txt_mlp_fc2", "ff_context_net_2")]
_SINGLE = [("single_blocks", "single_transformer_blocks"), ("linear1", "attn_to_QKVM"), ("linear2", "proj_out"),
           ("modulation_linear", "norm_linear")]


def convert_hunyuan_to_framepack(lora_sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """ref :160-234.  The down projection (and alpha) of a fused module is shared by its parts; the up projection
    is cut by rows into q | k | v (| mlp), q/k/v each as wide as the down projection's input."""
    out: Dict[str, Tensor] = {}
    for key, w in lora_sd.items():
        orig = key
        if "double_blocks" in key:
            table = _DOUBLE
        elif "single_blocks" in key:
            table = _SINGLE
        else:
            print(f"Unsupported module name: {key}, only double_blocks and single_blocks are supported")
            continue
        for a, b in table:
            key = key.replace(a, b)
        if "QKVM" in key:
            parts = [key.replace("QKVM", "q"), key.replace("QKVM", "k"), key.replace("QKVM", "v"),
                     key.replace("attn_to_QKVM", "proj_mlp")]
        elif "QKV" in key:
            parts = [key.replace("QKV", "q"), key.replace("QKV", "k"), key.replace("QKV", "v")]
        else:
            out[key] = w
            continue
        if "_down" in key or "alpha" in key:
            for p in parts:
                out[p] = w
        elif "_up" in key:
            # q, k and v are each as wide as the module's input = the columns of its own down matrix
            down = lora_sd.get(orig.replace("lora_up", "lora_down"))
            d = int(down.size(1)) if down is not None else (w.size(0) // 3 if len(parts) == 3 else w.size(0) // 7)
            assert w.size(0) >= 3 * d and (len(parts) == 4 or w.size(0) == 3 * d), f"fused weight size mismatch: {key}. {w.size()}"
            for i, p in enumerate(parts):
                out[p] = w[i * d:(i + 1) * d] if i < 3 else w[3 * d:]
        else:
            print(f"Unsupported module name: {key}")
    return out
