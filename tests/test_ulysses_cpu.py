"""Sequence-parallel exchange (framepack_amd.ulysses) on world_size-2 and -4 gloo groups on the CPU:
sharded attention == unsharded attention.  The attention itself is injected (the oracle's SDPA here; the
product injects the HIP kernel), so this covers the all-to-all index math, the padding rows at the end of the
sequence, and the row gathers by construction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from framepack_amd import ulysses
from oracle import dit_oracle as O


def _attn(q, k, v, heads, out):
    S = q.shape[0]
    o = O.sdpa_nhd(q.reshape(1, S, heads, 128), k.reshape(1, S, heads, 128), v.reshape(1, S, heads, 128))
    out.copy_(o.reshape(S, heads * 128))
    return out


def _worker(rank, world, port, n_img_total, n_txt, heads, strided, chunks, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(max(1, 8 // world))                  # (8 CPUs here: the ranks share them)
    try:
        d = heads * 128
        g = torch.Generator("cpu").manual_seed(0)
        q, k, v = (torch.randn(n_img_total + n_txt, d, generator=g) for _ in range(3))
        full = torch.empty_like(q)
        _attn(q, k, v, heads, full)
        S = n_img_total + n_txt
        n, lo, valid = ulysses.row_shard(S, world, rank)
        pad = torch.full((n - valid, d), float("nan"))          # padding rows: whatever they hold must not matter
        lq0, lk0, lv0 = (torch.cat([t[lo:lo + valid], pad]) for t in (q, k, v))
        if strided == "packed":                       # the send buffer already filled by the producer of q | k | v
            hl = heads // world                       # (layout contract of ops.qk_rmsnorm_rope_pack / norm_kernels.hip)
            hc = hl // chunks
            pack = torch.empty(chunks, world, n, 3 * hc * 128)
            for h in range(heads):
                peer, chunk, hi = h // hl, (h % hl) // hc, h % hc
                for which, t in enumerate((lq0, lk0, lv0)):
                    pack[chunk, peer, :, which * hc * 128 + hi * 128:which * hc * 128 + (hi + 1) * 128] = t[:, h * 128:(h + 1) * 128]
            raw = torch.full((n, d), float("nan"))    # q, k, v themselves are then only read for their shape
            got = ulysses.ulysses_attention(raw, raw, raw, S, heads, 128, _attn, chunks=chunks, packed=pack)
        elif strided:                                 # q|k|v as column ranges of one buffer, output over q
            buf = torch.cat([lq0, lk0, lv0, torch.zeros(n, 64)], dim=1)
            lq, lk, lv = buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:3 * d]
            got = ulysses.ulysses_attention(lq, lk, lv, S, heads, 128, _attn, out=lq, chunks=chunks)
        else:
            calls = []                                # caller's work under the exchange: run once, exchange asynchronous
            got = ulysses.ulysses_attention(lq0.contiguous(), lk0.contiguous(), lv0.contiguous(), S, heads, 128, _attn,
                                            chunks=chunks, overlap=lambda: calls.append(1))
            assert calls == [1]
        err = (got[:valid] - full[lo:lo + valid]).abs().max().item() if valid else 0.0
        gathered = ulysses.gather_rows(got.contiguous())[:S]
        err2 = (gathered - full).abs().max().item()
        ret[rank] = max(err, err2)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_img,n_txt,heads,strided,chunks", [
    (2, 64, 9, 2, False, 1), (2, 50, 0, 4, True, 1), (4, 40, 7, 4, True, 1), (4, 41, 2, 4, False, 1),
    (2, 63, 10, 2, True, 1), (4, 2, 1, 4, False, 1),
    (2, 63, 10, 4, True, 2), (2, 40, 7, 8, False, 2), (4, 41, 2, 8, True, 2),       # two head chunks, overlapped exchanges
    (2, 40, 7, 4, "packed", 1), (2, 63, 10, 8, "packed", 2), (4, 41, 2, 8, "packed", 2)])   # pre-filled send buffer
def test_ulysses_attention_equals_unsharded(world, n_img, n_txt, heads, strided, chunks):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), n_img, n_txt, heads, strided, chunks, ret), nprocs=world, join=True)
    assert len(ret) == world
    assert max(ret.values()) < 1e-5


def test_chunking_keeps_the_chip_full():
    """Two chunks only where half the local heads still give about a chip's worth of 256-row work items."""
    assert ulysses.pick_chunks(19385, 12) == 2 and ulysses.pick_chunks(19385, 6) == 2      # P = 2, 4
    assert ulysses.pick_chunks(19385, 3) == 1                                              # P = 8: 3 heads, odd
    assert ulysses.pick_chunks(1408, 2) == 1                                               # tiny sequences: one exchange


def test_row_shard():
    assert ulysses.row_shard(19385, 8, 0) == (2424, 0, 2424)
    assert ulysses.row_shard(19385, 8, 7) == (2424, 16968, 2417)
    assert ulysses.row_shard(3, 4, 2) == (1, 2, 1)
    assert ulysses.row_shard(3, 4, 3) == (1, 3, 0)            # a rank with only padding
    assert ulysses.row_shard(0, 4, 2) == (0, 0, 0)
