"""Small collectives used by the multi-GPU paths, written against torch.distributed so that they run on RCCL
(backend "nccl") on the GPUs and on gloo in the tests.  gloo moves CPU tensors only for point-to-point and
all-to-all traffic: GPU tensors on a gloo group (the two-processes-on-one-GPU tests) are staged through host memory;
RCCL takes the direct path."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def all_gather_padded(x: torch.Tensor, sizes: Sequence[int], group=None) -> List[torch.Tensor]:
    """All-gather tensors whose FIRST dimension differs per rank (sizes[r] on rank r): padded to the largest, gathered
    in one collective, cut back.  Returns the list of per-rank tensors in rank order."""
    world = dist.get_world_size(group)
    assert len(sizes) == world and x.shape[0] == sizes[dist.get_rank(group)]
    n = max(sizes)
    pad = x if x.shape[0] == n else torch.cat([x, x.new_zeros((n - x.shape[0],) + tuple(x.shape[1:]))], dim=0)
    pad = pad.contiguous()
    if _staged(x, group):
        out = torch.empty((world * n,) + tuple(x.shape[1:]), dtype=x.dtype)
        dist.all_gather_into_tensor(out, pad.cpu(), group=group)
        out = out.to(x.device)
    else:
        out = torch.empty((world * n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * n:r * n + sizes[r]] for r in range(world)]


class _Halo:
    """Handle of a posted halo exchange: `wait()` -> (from_up, from_down)."""

    def __init__(self, works, bufs, like_up, like_down):
        self._works, self._bufs, self._like = works, bufs, (like_up, like_down)

    def wait(self) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        for w in self._works:
            w.wait()                      # RCCL: the current stream waits for the communicator's; gloo: the host does
        self._works = []

        def back(key, like):
            if like is None:
                return None
            b = self._bufs[key]
            return b.to(like.device) if b.device != like.device else b
        return back("up", self._like[0]), back("down", self._like[1])


def neighbour_exchange_start(send_up: Optional[torch.Tensor], send_down: Optional[torch.Tensor], group=None,
                             cache: Optional[dict] = None) -> _Halo:
    """Post the halo exchange along a chain of ranks and return at once: rank r sends `send_up` to rank r - 1 and
    `send_down` to rank r + 1 (None at the ends of the chain) and receives tensors of the same shapes from those
    neighbours.  All transfers are posted in one batch; what the caller launches before `wait()` runs while they travel
    (on RCCL the transfers are on the communicator's stream).  cache: a dict the caller keeps - send staging and receive
    buffers are then allocated once per shape instead of once per call (a buffer is free again when the caller has
    consumed what `wait()` returned, which is before it posts the next exchange of that shape)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ops, bufs = [], {}

    def peer(r):
        return dist.get_global_rank(group, r) if group is not None else r

    def buf(kind, key, like):
        if cache is None:
            return torch.empty_like(like)
        k = (kind, key, tuple(like.shape), like.dtype, str(like.device))
        b = cache.get(k)
        if b is None:
            b = cache[k] = torch.empty_like(like)
        return b
    for key, t, r in (("up", send_up, rank - 1), ("down", send_down, rank + 1)):
        if t is None:
            continue
        assert 0 <= r < world
        if _staged(t, group):
            src = t.contiguous().cpu()
        elif t.is_contiguous():
            src = t
        else:
            src = buf("send", key, torch.empty(t.shape, dtype=t.dtype, device=t.device))
            src.copy_(t)
        dst = buf("recv", key, src)
        bufs[key] = dst
        ops.append(dist.P2POp(dist.isend, src, peer(r), group))
        ops.append(dist.P2POp(dist.irecv, dst, peer(r), group))
    works = dist.batch_isend_irecv(ops) if ops else []
    return _Halo(works, bufs, send_up, send_down)


def neighbour_exchange(send_up: Optional[torch.Tensor], send_down: Optional[torch.Tensor],
                       group=None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """`neighbour_exchange_start(...).wait()`: the blocking form."""
    return neighbour_exchange_start(send_up, send_down, group).wait()
