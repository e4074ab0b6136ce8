"""Weight formats around the DiT: LoRA merge into a FramePack state dict, and the FP8
(E4M3 weight + `scale_weight`) checkpoint optimisation the reference applies at load time.

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

FP8, with the reference's names and arguments (ref utils/fp8_optimization_utils.py; call sites framepack/models.py:255-275):
    calculate_fp8_maxval, quantize_tensor_to_fp8                               ref :8-81
    optimize_state_dict_with_fp8(state_dict, calc_device, targets, excludes)   ref :84-167
    apply_fp8_monkey_patch(model, optimized_state_dict)                        ref :229-277
The result is the reference's own layout - `<linear>.weight` in float8_e4m3fn plus `<linear>.scale_weight` [1] in the
weight's original dtype - which HunyuanVideoTransformer3DModelPacked consumes as it stands (model.py `_dense_weight`
for the W8A16 semantics of ref :216-226, `_fuse_fp8` for the fp8 MFMA path).  Every intermediate is evaluated in the
weight's own dtype and in the reference's operation order, so bytes and scales are identical (tests/golden/fp8.npz).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional, Union

import torch
from torch import nn

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


# ------------------------------------------------------------------------------------------------ FP8 checkpoints
def calculate_fp8_maxval(exp_bits: int = 4, mantissa_bits: int = 3, sign_bits: int = 1) -> float:
    """Largest magnitude the reference attributes to an FP8 format (ref :8-34): 448 for E4M3; for E5M2 its formula
    gives 98304 (it counts the all-ones exponent as finite), which is kept."""
    if exp_bits + mantissa_bits + sign_bits != 8:
        raise AssertionError("Total bits must be 8")
    bias = 2 ** (exp_bits - 1) - 1
    top_mantissa = 1.0 + sum(2.0 ** -(i + 1) for i in range(mantissa_bits - 1))
    return top_mantissa * (2 ** (2 ** exp_bits - 1 - bias))


def quantize_tensor_to_fp8(tensor: Tensor, scale, exp_bits: int = 4, mantissa_bits: int = 3, sign_bits: int = 1,
                           max_value: Optional[float] = None, min_value: Optional[float] = None):
    """`tensor / scale` clamped to the format's range and rounded onto its grid, still in `tensor`'s dtype
    (ref :37-81).  Returns (values on the grid, scale)."""
    bias = 2 ** (exp_bits - 1) - 1
    if max_value is None:
        max_value = calculate_fp8_maxval(exp_bits, mantissa_bits, sign_bits)
        min_value = -max_value if sign_bits > 0 else 0.0
    clamped = torch.clamp(tensor / scale, min_value, max_value)
    magnitude = clamped.abs()
    # biased exponent of every non-zero element (zero elsewhere), not below the first normal binade
    exponent = torch.where(magnitude > 0, torch.floor(torch.log2(magnitude) + bias), torch.zeros_like(clamped))
    step = 2.0 ** (torch.clamp(exponent, min=1.0) - mantissa_bits - bias)
    return torch.round(clamped / step) * step, scale


def _fp8_targets(keys: Iterable[str], include, exclude):
    for key in keys:
        if not key.endswith(".weight"):
            continue
        if include is not None and not any(p in key for p in include):
            continue
        if exclude is not None and any(p in key for p in exclude):
            continue
        yield key


def optimize_state_dict_with_fp8(state_dict: Dict[str, Tensor], calc_device, target_layer_keys=None,
                                 exclude_layer_keys=None, exp_bits: int = 4, mantissa_bits: int = 3,
                                 move_to_device: bool = False) -> Dict[str, Tensor]:
    """Replace, in place, every targeted `.weight` by its FP8 image and add `<module>.scale_weight` = max|W| / maxval,
    one scale per tensor, shape [1], in W's dtype (ref :84-167; the product passes targets
    ["transformer_blocks", "single_transformer_blocks"] and excludes ["norm"], ref framepack/models.py:259-260)."""
    formats = {(4, 3): torch.float8_e4m3fn, (5, 2): torch.float8_e5m2}
    if (exp_bits, mantissa_bits) not in formats:
        raise ValueError(f"Unsupported FP8 format: E{exp_bits}M{mantissa_bits}")
    fp8 = formats[(exp_bits, mantissa_bits)]
    top = calculate_fp8_maxval(exp_bits, mantissa_bits)
    picked = [k for k in _fp8_targets(list(state_dict.keys()), target_layer_keys, exclude_layer_keys)
              if isinstance(state_dict[k], Tensor)]
    for key in picked:
        w = state_dict[key]
        home, dtype = w.device, w.dtype
        if calc_device is not None:
            w = w.to(calc_device)
        scale = torch.max(torch.abs(w.flatten())) / top
        q, _ = quantize_tensor_to_fp8(w, scale, exp_bits, mantissa_bits, 1, top, -top)
        q = q.to(fp8)
        if not move_to_device:
            q = q.to(home)
        state_dict[key] = q
        state_dict[key[: -len(".weight")] + ".scale_weight"] = torch.tensor([scale], dtype=dtype, device=q.device)
    print(f"Number of optimized Linear layers: {len(picked)}")
    return state_dict


def apply_fp8_monkey_patch(model: nn.Module, optimized_state_dict: Dict[str, Tensor], use_scaled_mm: bool = False):
    """Prepare `model` for `load_state_dict(optimized_state_dict, strict=True, assign=True)`: every nn.Linear that
    has a `.scale_weight` entry gets that buffer (ref :229-277).  The reference also swaps the layer's forward for a
    dequantise-then-linear; here the packed forward reads `scale_weight` itself when it fuses the block's weights
    (model.py `_dense_weight`), so no method is replaced.  `use_scaled_mm` (torch._scaled_mm on the reference's
    side) maps to `model.enable_fp8_compute()`, the fp8 MFMA path."""
    paths = {k[: -len(".scale_weight")] for k in optimized_state_dict if k.endswith(".scale_weight")}
    patched = 0
    for name, module in model.named_modules():
        if name in paths and isinstance(module, nn.Linear):
            module.register_buffer("scale_weight", torch.tensor(1.0, dtype=module.weight.dtype))
            patched += 1
    if use_scaled_mm and hasattr(model, "enable_fp8_compute"):
        model.enable_fp8_compute(True)
    print(f"Number of monkey-patched Linear layers: {patched}")
    return model


# ------------------------------------------------------------------------------------------------ load sequence
FP8_TARGET_KEYS = ["transformer_blocks", "single_transformer_blocks"]      # ref framepack/models.py:259
FP8_EXCLUDE_KEYS = ["norm"]                                                 # ref framepack/models.py:260


def load_transformer(model_dir: str, lora_file: Optional[str] = None, lora_multiplier: float = 0.8,
                     fp8_optimization: bool = False, calc_device=None, torch_dtype=torch.bfloat16):
    """The reference's transformer load sequence (ref framepack/models.py:233-275) on this package's model class:
    bf16 weights from a local directory, then - on the state dict - LoRA merge first and FP8 optimisation second,
    then an assigning strict load.  A LoRA path that does not exist is skipped with the reference's warning.
    `calc_device`: where the merge / quantisation arithmetic runs (the reference uses the GPU); None = in place."""
    from .model import HunyuanVideoTransformer3DModelPacked
    transformer = HunyuanVideoTransformer3DModelPacked.from_pretrained(model_dir, torch_dtype=torch_dtype).cpu()
    transformer.eval()
    transformer.high_quality_fp32_output_for_inference = True
    if lora_file is None and not fp8_optimization:
        return transformer
    state_dict = transformer.state_dict()
    if lora_file is not None and os.path.exists(lora_file):
        print(f"Merging LoRA file {os.path.basename(lora_file)} with multiplier {lora_multiplier}...")
        device = torch.device(calc_device) if calc_device is not None else torch.device("cpu")
        state_dict = merge_lora_to_state_dict(state_dict, lora_file, lora_multiplier, device)
    elif lora_file is not None:
        print(f"Warning: LoRA file {lora_file} not found, skipping LoRA application")
    if fp8_optimization:
        state_dict = optimize_state_dict_with_fp8(state_dict, calc_device, FP8_TARGET_KEYS, FP8_EXCLUDE_KEYS,
                                                  move_to_device=False)
        apply_fp8_monkey_patch(transformer, state_dict, use_scaled_mm=False)
    info = transformer.load_state_dict(state_dict, strict=True, assign=True)
    print(f"Model modifications applied: {info}")
    return transformer
