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

import os
from typing import Callable, Optional

import torch
import torch.distributed as dist

# FRAMEPACK_SP_ASYNC=0: every exchange is issued synchronously on the current stream's order (no `async_op`, nothing runs
# under an exchange) - the fall-back if the overlapped RCCL path misbehaves on a machine; results are the same.
SP_ASYNC = os.environ.get("FRAMEPACK_SP_ASYNC", "1") != "0"


def row_shard(n_rows: int, world: int, rank: int):
    """(rows per rank n, first global row, valid rows) of this rank's run of an n_rows-long sequence."""
    n = -(-n_rows // world) if n_rows else 0
    lo = min(rank * n, n_rows)
    return n, lo, min(n, n_rows - lo)


class _Done:
    def wait(self):
        return None


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, group, async_op: bool = False):
    """all_to_all_single -> a handle with .wait().  gloo has no GPU all-to-all, so GPU tensors on a gloo group (the
    2-process test on one GPU) are staged through host memory, synchronously.  RCCL takes the direct path; with
    async_op the exchange runs on the communicator's stream and `wait()` makes the current stream wait for it."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()
    async_op = async_op and SP_ASYNC
    w = dist.all_to_all_single(out, inp, group=group, async_op=async_op)
    return w if async_op else _Done()


_CUS = {}


def _cu_count(device=None) -> int:
    """Compute units of the device the exchange runs on (256 on an MI355X); 256 where there is no GPU (CPU tests)."""
    if not torch.cuda.is_available():
        return 256
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    if idx not in _CUS:
        _CUS[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _CUS[idx]


def pick_chunks(n_total: int, heads_local: int, cus: Optional[int] = None) -> int:
    """Two head chunks (the second chunk's exchange runs under the first chunk's attention, the first chunk's return
    under the second's attention) when each chunk still fills the chip: a chunk has heads_local / 2 heads x
    ceil(S / 256) work items of 256 query rows, and a partly filled round costs a whole sweep.  At S = 19 385 on 256 CUs
    that is P = 2 and 4 (6 and 3 heads per chunk: 456 and 228 items); at P = 8 the 3 local heads stay one exchange."""
    if cus is None:
        cus = _cu_count()
    if heads_local % 2 == 0 and (heads_local // 2) * -(-n_total // 256) >= (3 * cus) // 4:
        return 2
    return 1


def exchange_buffers(cache: Optional[dict], chunks: int, world: int, n: int, hwc: int, dtype, device):
    """The exchange's four buffers - send [chunks, world, n, 3 hwc], joint [chunks, world n, 3 hwc], back [chunks,
    world n, hwc], recv [chunks, world, n, hwc] - from `cache` (allocated once per shape: 60 attentions per step reuse
    them; a buffer an exchange still reads or writes is never touched before that exchange's handle has been waited
    for, see ulysses_attention) or fresh when no cache is given."""
    key = (chunks, world, n, hwc, dtype, str(device))
    b = None if cache is None else cache.get(key)
    if b is None:
        b = dict(send=torch.empty((chunks, world, n, 3 * hwc), dtype=dtype, device=device),
                 joint=torch.empty((chunks, world * n, 3 * hwc), dtype=dtype, device=device),
                 back=torch.empty((chunks, world * n, hwc), dtype=dtype, device=device),
                 recv=torch.empty((chunks, world, n, hwc), dtype=dtype, device=device))
        if cache is not None:
            cache.clear()                   # one shape at a time (a negative prompt's other S replaces it)
            cache[key] = b
    return b


def ulysses_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_total: int, heads: int, head_dim: int,
                      attn_fn: Callable, group=None, out: Optional[torch.Tensor] = None,
                      chunks: Optional[int] = None, packed: Optional[torch.Tensor] = None,
                      overlap: Optional[Callable[[], None]] = None, bufs: Optional[dict] = None) -> torch.Tensor:
    """q, k, v: [n, heads*head_dim] views (row-strided ok) holding this rank's run of the packed sequence (row_shard;
    padding rows included).  n_total = S, the valid rows over all ranks.  Returns / fills `out` (same shape; may be q)
    with the attention output of the local rows for all heads (padding rows: zeros).

    attn_fn(q, k, v, heads_local, out) -> out   over [S, heads_local*head_dim] views.
    chunks: head chunks exchanged and attended one after the other so that communication overlaps attention
    (default: `pick_chunks`).
    packed: the send buffer [chunks, world, n, 3 * hc * head_dim] already filled by the producer of q | k | v
    (ops.qk_rmsnorm_rope_pack: the RMSNorm/RoPE kernel writes it directly); q, k, v are then only read for their shape.
    overlap: work of the caller that needs nothing from the attention (the single blocks' MLP in-projection); it is
    called once, after the forward exchanges have been issued - asynchronously then, whatever `chunks` - and before
    the first of them is waited for, so its kernels run while q | k | v travel.
    bufs: a dict the caller keeps (one per model): the exchange buffers are then allocated once per shape instead of
    once per attention (`exchange_buffers`).  Safe to reuse from one attention to the next: a collective starts behind
    everything queued on the current stream when it is issued (the previous attention's reads and unpack copies), and
    every handle is waited for before the current stream touches what that collective reads or writes."""
    world = dist.get_world_size(group)
    if heads % world != 0:
        raise ValueError(f"{heads} heads do not divide over {world} ranks")
    hl = heads // world
    n = q.shape[0]
    if n != row_shard(n_total, world, 0)[0]:
        raise ValueError(f"{n} local rows do not match {n_total} rows over {world} ranks")
    if chunks is None:
        chunks = pick_chunks(n_total, hl)
    if hl % chunks != 0:
        raise ValueError(f"{hl} local heads do not split into {chunks} chunks")
    hc = hl // chunks
    hwc = hc * head_dim
    dev, dt = q.device, q.dtype
    S = n_total

    # ---- pack: [n, P, chunk, (q|k|v), hwc] -> [chunk, P, n, 3*hwc]: per chunk one contiguous slice per peer; what a
    #      peer receives is the packed sequence in order (rank order = sequence order)
    B = exchange_buffers(bufs, chunks, world, n, hwc, dt, dev)
    if packed is not None:
        if tuple(packed.shape) != (chunks, world, n, 3 * hwc) or not packed.is_contiguous():
            raise ValueError(f"packed send buffer {tuple(packed.shape)} does not match {(chunks, world, n, 3 * hwc)}")
        send = packed
    else:
        send = B["send"]
        for i, t in enumerate((q, k, v)):
            send.view(chunks, world, n, 3, hwc)[:, :, :, i].copy_(t.reshape(n, world, chunks, hwc).permute(2, 1, 0, 3))
    joints = B["joint"]
    fwd = [_all_to_all(joints[c], send[c].view(world * n, 3 * hwc), group, async_op=chunks > 1 or overlap is not None)
           for c in range(chunks)]
    if overlap is not None:
        overlap()

    # ---- per chunk: attention over the valid tokens of the local head chunk, written straight into the send buffer
    #      of the inverse exchange (rows past S: padding, zeros), which then runs under the next chunk's attention
    inv = []
    for c in range(chunks):
        fwd[c].wait()
        back = B["back"][c]
        attn_fn(joints[c][:S, :hwc], joints[c][:S, hwc:2 * hwc], joints[c][:S, 2 * hwc:], hc, back[:S])
        if S < world * n:
            back[S:].zero_()
        inv.append(_all_to_all(B["recv"][c].view(world * n, hwc), back, group, async_op=chunks > 1))

    if out is None:
        out = torch.empty((n, heads * head_dim), dtype=dt, device=dev)
    # recv[c][p] = heads [p*hl + c*hc, +hc) of my rows  ->  out[:, (p*chunks + c)*hwc : +hwc]: one strided copy per chunk
    # (out may be a row-strided view - the q columns of the projection workspace)
    ov = out.unflatten(1, (world, chunks, hwc))
    for c in range(chunks):
        inv[c].wait()
        ov[:, :, c].copy_(B["recv"][c].permute(1, 0, 2))
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
