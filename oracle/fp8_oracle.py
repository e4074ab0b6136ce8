"""CPU restatement of the reference's FP8 (E4M3) weight quantisation.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `ref:` = /root/reference/utils/fp8_optimization_utils.py.
Pinned by tests/golden/fp8.npz (made from that importable module).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

Tensor = torch.Tensor


def fp8_maxval(exp_bits: int = 4, mantissa_bits: int = 3) -> float:
    """ref :8-34.  448 for E4M3 (and 98304 for E5M2, the reference's value, not IEEE's 57344)."""
    bias = 2 ** (exp_bits - 1) - 1
    mant = 1.0 + sum(2.0 ** -(i + 1) for i in range(mantissa_bits - 1))
    return mant * (2 ** (2 ** exp_bits - 1 - bias))


def round_to_fp8_grid(t: Tensor, scale, exp_bits: int = 4, mantissa_bits: int = 3) -> Tensor:
    """ref :37-81 - clamp(t/scale) then round to the E4M3 grid (still in t's dtype)."""
    mx = fp8_maxval(exp_bits, mantissa_bits)
    bias = 2 ** (exp_bits - 1) - 1
    c = torch.clamp(t / scale, -mx, mx)
    a = c.abs()
    nz = a > 0
    log_scales = torch.zeros_like(c)
    if nz.any():
        log_scales[nz] = torch.floor(torch.log2(a[nz]) + bias)
    log_scales = torch.clamp(log_scales, min=1.0)
    q = 2.0 ** (log_scales - mantissa_bits - bias)
    return torch.round(c / q) * q


def is_fp8_target(key: str, targets: Optional[Iterable[str]], excludes: Optional[Iterable[str]]) -> bool:
    """ref :118-125."""
    hit = (targets is None or any(p in key for p in targets)) and key.endswith(".weight")
    return hit and not (excludes is not None and any(p in key for p in excludes))


def quantize_state_dict(sd: Dict[str, Tensor], targets=("transformer_blocks", "single_transformer_blocks"),
                        excludes=("norm",)) -> Dict[str, Tensor]:
    """ref :84-167 with the product's key filters (framepack/models.py:259-260).
    scale = max|W| / 448 per tensor, stored as `<name>.scale_weight` [1] in W's original dtype."""
    mx = fp8_maxval()
    out = dict(sd)
    for key in [k for k in sd if is_fp8_target(k, targets, excludes)]:
        w = sd[key]
        scale = torch.max(torch.abs(w.flatten())) / mx
        out[key] = round_to_fp8_grid(w, scale).to(torch.float8_e4m3fn)
        out[key.replace(".weight", ".scale_weight")] = torch.tensor([scale], dtype=w.dtype)
    return out


def fp8_linear(x: Tensor, w8: Tensor, scale: Tensor, bias: Optional[Tensor]) -> Tensor:
    """ref :216-226 - the shipped W8A16 path: dequantise then F.linear."""
    return torch.nn.functional.linear(x, w8.to(scale.dtype) * scale, bias)
