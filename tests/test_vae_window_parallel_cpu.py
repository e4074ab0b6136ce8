"""Window-parallel VAE decode (framepack_amd.vae.AutoencoderKLHunyuanVideo._framewise) on a world_size-2/3 gloo
group on the CPU: windows decoded by different ranks and broadcast == the oracle's single-process framewise
decode.  The per-window decoder is injected (the oracle's decode_tile here; the product injects the HIP decoder),
so this covers the window/ownership/broadcast/cross-fade logic by construction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from framepack_amd.vae import AutoencoderKLHunyuanVideo
from oracle import vae_oracle as VO

CFG = dict(block_out_channels=(32, 32, 64, 64))


def _worker(rank, world, port, T, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(max(1, 8 // world))                  # (8 CPUs here: the ranks share them)
    try:
        cfg = VO.VAEConfig(**CFG)
        P = VO.random_state_dict(cfg, seed=3)
        z = torch.randn(1, 16, T, 2, 2, generator=torch.Generator().manual_seed(7))
        want = VO.decode(P, cfg, z, framewise=True)
        vae = AutoencoderKLHunyuanVideo(**CFG)
        calls = []

        def tile_fn(zt):
            calls.append(zt.shape[2])
            return VO.decode_tile(P, cfg, zt)

        single = vae._framewise(z, tile_fn)                      # no group: every window local
        n_windows = len(calls)
        calls.clear()
        vae.enable_window_parallel()
        got = vae._framewise(z, tile_fn)
        ret[rank] = (float((got - want).abs().max()), float((single - want).abs().max()), len(calls), n_windows)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,T", [(2, 9), (3, 7), (2, 5)])
def test_window_parallel_decode_equals_single_process(world, T):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), T, ret), nprocs=world, join=True)
    assert len(ret) == world
    n_windows = ret[0][3]
    assert n_windows == len(range(0, T, 3))
    assert sum(r[2] for r in ret.values()) == n_windows          # every window decoded exactly once
    for err, err_single, _, _ in ret.values():
        assert err == 0.0 and err_single == 0.0
