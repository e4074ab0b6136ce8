"""Ulysses-style sequence parallelism for the joint attention of the packed DiT (RCCL all-to-all over xGMI).

New design (the reference is single-GPU: SURVEY.md 2a); the contract is "same result as 1 GPU".
  * the PACKED sequence [image rows | image-projection + text rows] (S rows) is cut into P equal runs of
    n = ceil(S / P) rows: rank r holds rows [r n, (r+1) n); the run that crosses the image/text boundary simply holds
    both kinds, and the rows past S (on the last rank(s)) are padding - computed, never attended to, never returned.
    Every token-wise op (LayerNorm+modulate, GEMMs, RMSNorm+RoPE, gated residual) runs on the n local rows: 1/P of
    the single-GPU work for any S and P (no divisibility condition on the image token count; the first version
    replicated the text rows: +35 % GEMM rows per rank at P = 8);
  * attention needs all tokens of a head: one all-to-all turns [n rows, all 24 heads] into [P n rows, 24/P heads]
    for q, k and v together (one exchange, 3 d n elements per rank); rank order = sequence order, so the received
    buffer IS the packed sequence with the padding at its end and the K1 kernel runs on its first S rows; the
    inverse all-to-all returns [n rows, all heads].
xGMI is point-to-point: an all-to-all sends a distinct slice to each of the 7 peers and uses all links at once.

The exchange is written against torch.distributed so that it runs on RCCL (backend "nccl") on the GPUs and on
gloo in the CPU tests; the attention itself is injected (`attn_fn`): the product passes framepack_amd.ops.attention.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def row_shard(n_rows: int, world: int, rank: int):
    """(rows per rank n, first global row, valid rows) of this rank's run of an n_rows-long sequence."""
    n = -(-n_rows // world) if n_rows else 0
    lo = min(rank * n, n_rows)
    return n, lo, min(n, n_rows - lo)


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    """all_to_all_single; gloo has no GPU all-to-all, so GPU tensors on a gloo group (the 2-process test on one GPU)
    are staged through host memory.  RCCL takes the direct path."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, group=group)


def ulysses_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_total: int, heads: int, head_dim: int,
                      attn_fn: Callable, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q, k, v: [n, heads*head_dim] views (row-strided ok) holding this rank's run of the packed sequence (row_shard;
    padding rows included).  n_total = S, the valid rows over all ranks.  Returns / fills `out` (same shape; may be q)
    with the attention output of the local rows for all heads (padding rows: zeros).

    attn_fn(q, k, v, heads_local, out) -> out   over [S, heads_local*head_dim] views."""
    world = dist.get_world_size(group)
    if heads % world != 0:
        raise ValueError(f"{heads} heads do not divide over {world} ranks")
    hl = heads // world
    hw = hl * head_dim
    n = q.shape[0]
    if n != row_shard(n_total, world, 0)[0]:
        raise ValueError(f"{n} local rows do not match {n_total} rows over {world} ranks")
    dev, dt = q.device, q.dtype

    # ---- forward exchange: [n, P, (q|k|v), hw] -> per-peer contiguous [P, n, 3*hw]; received = the sequence in order
    send = torch.stack([q.reshape(n, world, hw), k.reshape(n, world, hw), v.reshape(n, world, hw)],
                       dim=2).permute(1, 0, 2, 3).contiguous()
    joint = torch.empty((world * n, 3 * hw), dtype=dt, device=dev)
    _all_to_all(joint, send.view(world * n, 3 * hw), group)

    # ---- attention over the valid tokens, local head group (output over q)
    S = n_total
    o = attn_fn(joint[:S, :hw], joint[:S, hw:2 * hw], joint[:S, 2 * hw:], hl, joint[:S, :hw])

    # ---- inverse exchange: to peer p its rows, my head group
    back = torch.empty((world * n, hw), dtype=dt, device=dev)
    back[:S] = o
    if S < world * n:
        back[S:].zero_()
    recv = torch.empty((world, n, hw), dtype=dt, device=dev)
    _all_to_all(recv.view(world * n, hw), back, group)
    if out is None:
        out = torch.empty((n, heads * head_dim), dtype=dt, device=dev)
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
    if x.is_cuda and dist.get_backend(group) == "gloo":          # see _all_to_all
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, x.contiguous().cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out
