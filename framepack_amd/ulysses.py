"""Ulysses-style sequence parallelism for the joint attention of the packed DiT (RCCL all-to-all over xGMI).

New design (the reference is single-GPU: SURVEY.md 2a); the contract is "same result as 1 GPU".
  * the IMAGE token rows are sharded evenly over the P ranks (n = Si / P rows each); every token-wise op
    (LayerNorm+modulate, GEMMs, RMSNorm+RoPE, gated residual) runs on the local rows only;
  * the TEXT rows (<= 985, ~5 % of the sequence, data-dependent length) are replicated: every rank computes
    them for all heads, so their length never has to be split;
  * attention needs all tokens of a head: one all-to-all turns [n rows, all 24 heads] into
    [all Si rows, 24/P heads] for q, k and v together (one exchange, 3*d*n elements per rank), the local
    rank's head group of the replicated text rows is appended, the K1 kernel runs on 24/P heads, and the
    inverse all-to-all returns [n rows, all heads]; the text rows of the output ride in the same exchange
    (every peer gets them), so no second collective is needed.
xGMI is point-to-point: an all-to-all sends a distinct slice to each of the 7 peers and uses all links at once.

The exchange is written against torch.distributed so that it runs on RCCL (backend "nccl") on the GPUs and on
gloo in the CPU tests; the attention itself is injected (`attn_fn`): the product passes framepack_amd.ops.attention.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int):
    """Even row split; n_rows must be a multiple of world (the caller pads the image stream otherwise)."""
    if n_rows % world != 0:
        raise ValueError(f"image token count {n_rows} is not a multiple of the sequence-parallel degree {world}")
    n = n_rows // world
    return rank * n, (rank + 1) * n


def ulysses_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_img: int, heads: int, head_dim: int,
                      attn_fn: Callable, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q, k, v: [n_img + n_txt, heads*head_dim] views (row-strided ok) holding this rank's image rows followed by
    the replicated text rows.  Returns / fills `out` (same shape; may be q) with the attention output of the
    local image rows and of all text rows, for all heads.

    attn_fn(q, k, v, heads_local, out) -> out   over [Si + n_txt, heads_local*head_dim] views."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if heads % world != 0:
        raise ValueError(f"{heads} heads do not divide over {world} ranks")
    hl = heads // world
    hw = hl * head_dim
    rows = q.shape[0]
    n_txt = rows - n_img
    Si = n_img * world
    dev, dt = q.device, q.dtype

    # ---- forward exchange: [n_img, P, (q|k|v), hw] -> per-peer contiguous [P, n_img, 3*hw]
    send = torch.stack([q[:n_img].reshape(n_img, world, hw), k[:n_img].reshape(n_img, world, hw),
                        v[:n_img].reshape(n_img, world, hw)], dim=2).permute(1, 0, 2, 3).contiguous()
    joint = torch.empty((Si + n_txt, 3 * hw), dtype=dt, device=dev)
    dist.all_to_all_single(joint[:Si], send.view(world * n_img, 3 * hw), group=group)
    if n_txt:
        c0 = rank * hw
        joint[Si:, :hw] = q[n_img:, c0:c0 + hw]
        joint[Si:, hw:2 * hw] = k[n_img:, c0:c0 + hw]
        joint[Si:, 2 * hw:] = v[n_img:, c0:c0 + hw]

    # ---- attention over all tokens, local head group (output over q)
    jq, jk, jv = joint[:, :hw], joint[:, hw:2 * hw], joint[:, 2 * hw:]
    o = attn_fn(jq, jk, jv, hl, jq)

    # ---- inverse exchange: to peer p its image rows (+ all text rows), my head group
    back = torch.empty((world, n_img + n_txt, hw), dtype=dt, device=dev)
    back[:, :n_img] = o[:Si].reshape(world, n_img, hw)
    if n_txt:
        back[:, n_img:] = o[Si:].unsqueeze(0)
    recv = torch.empty_like(back)
    dist.all_to_all_single(recv.view(world * (n_img + n_txt), hw), back.view(world * (n_img + n_txt), hw), group=group)
    if out is None:
        out = torch.empty((rows, heads * head_dim), dtype=dt, device=dev)
    # recv[p] = head group p of my rows  ->  out[:, p*hw:(p+1)*hw]
    out_v = out.unflatten(1, (world, hw)) if out.is_contiguous() else None
    if out_v is not None:
        out_v.copy_(recv.permute(1, 0, 2))
    else:
        for p in range(world):
            out[:, p * hw:(p + 1) * hw] = recv[p]
    return out


def gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """all-gather equal row shards [n, c] -> [P*n, c] in rank order (output head, TeaCache residuals)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out
