"""Spatially tiled VAE decode with halo exchange (framepack_amd.vae.AutoencoderKLHunyuanVideo.enable_tile_parallel)
on world_size-2/3 gloo groups on the CPU: one band of rows per rank, border rows exchanged before every convolution,
GroupNorm statistics all-reduced, mid-block keys/values all-gathered == the same decoder run by one process.

The kernels are torch stand-ins with the signatures of framepack_amd.ops (tests/vae_torch_kernels.py, test
infrastructure): what is under test here is everything AROUND the kernels - band bookkeeping at every resolution, what
is sent to whom, the crop of halo-only rows, the statistics merge, the token order of the gathered keys, the frame-causal
mask per band, the final stitch.  The same comparison with the HIP kernels runs on the GPU (test_vae_tile_parallel_gpu.py)."""
import math
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CFG = dict(block_out_channels=(64, 64, 64, 64), norm_num_groups=16)


def _worker(rank, world, port, T, H, Wd, framewise, ret, hybrid=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(max(1, 8 // world))                  # (8 CPUs here: the ranks share them)
    try:
        import vae_torch_kernels as K
        from framepack_amd import vae as V
        from oracle import vae_oracle as VO
        V.ops = K                                              # the multi-rank logic over stand-in kernels
        cfg = VO.VAEConfig(**CFG)
        P = {k: v.half() for k, v in VO.random_state_dict(cfg, seed=3).items()}
        vae = V.AutoencoderKLHunyuanVideo(**CFG)
        vae.load_state_dict(P, strict=True)
        vae = vae.half().eval()
        z = torch.randn(1, 16, T, H, Wd, generator=torch.Generator().manual_seed(7)).half()
        run = (lambda: vae._framewise(z, vae._decode_tile)) if framewise else (lambda: vae._decode_tile(z))
        single = run()
        if hybrid:
            vae.enable_hybrid_parallel(hybrid)                 # windows dealt to groups of ranks, bands inside a group
        else:
            vae.enable_tile_parallel()
        tiled = run()
        # the overlapped exchange (border rows recomputed from three-row slabs) against the exchange-then-compute form:
        # the same operands per output voxel (the HIP kernel then gives the same bits, checked on the GPU; torch's CPU
        # convolution sums in an order that depends on the volume's shape)
        same = True
        if not hybrid:
            V._HALO_OVERLAP = False
            same = (run().float() - tiled.float()).abs().max().item() <= 4e-3
            V._HALO_OVERLAP = True
        want = VO.vae_decode({k: v.float() for k, v in P.items()}, cfg, z.float() * cfg.scaling_factor,
                             framewise=framewise) if rank == 0 else None
        mse = ((tiled.double() - single.double()) ** 2).mean().item()
        ret[rank] = dict(shape=tuple(tiled.shape), max_diff=(tiled.float() - single.float()).abs().max().item(), same=same,
                         psnr=10 * math.log10(4.0 / max(mse, 1e-30)),
                         psnr_oracle=None if want is None else
                         10 * math.log10(4.0 / max(((single.double() - want.double()) ** 2).mean().item(), 1e-30)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,T,H,W,framewise", [(2, 3, 4, 4, False), (3, 2, 7, 3, False), (2, 1, 2, 5, False),
                                                   (2, 6, 4, 4, True)])
def test_tiled_decode_with_halo_exchange_equals_single_process(world, T, H, W, framewise):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), T, H, W, framewise, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in ret.values():
        assert r["shape"] == (1, 3, 4 * (T - 1) + 1, 8 * H, 8 * W)
        # identical up to the rounding of the merged GroupNorm moments (fp16 volumes: one ulp at 1.0 is 1e-3)
        assert r["max_diff"] <= 1e-2 and r["psnr"] >= 60.0, r      # a few fp16 ulps at |x| ~ 2
        assert r["same"], r
    assert ret[0]["psnr_oracle"] >= 40.0, ret[0]        # and the stand-in decoder is the oracle's decoder


def test_windows_dealt_to_groups_of_band_ranks_equals_single_process():
    """enable_hybrid_parallel: 4 ranks = 2 window groups x 2 bands, 3 framewise windows (T = 7 latent frames)."""
    world, T, H, W = 4, 7, 2, 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), T, H, W, True, ret, 2), nprocs=world, join=True)
    assert len(ret) == world
    for r in ret.values():
        assert r["shape"] == (1, 3, 4 * (T - 1) + 1, 8 * H, 8 * W)
        assert r["max_diff"] <= 1e-2 and r["psnr"] >= 60.0 and r["same"], r


def test_window_and_tile_parallel_on_one_group_are_refused():
    """both modes on the same group would hang (a window's owner waits in the halo exchange for ranks that
    wait in the window's broadcast)."""
    import unittest.mock as um
    from framepack_amd.vae import AutoencoderKLHunyuanVideo
    vae = AutoencoderKLHunyuanVideo(**CFG)
    with um.patch("torch.distributed.group") as g:
        g.WORLD = object()
        vae.enable_tile_parallel()
        with pytest.raises(RuntimeError):
            vae.enable_window_parallel()
    vae2 = AutoencoderKLHunyuanVideo(**CFG)
    with um.patch("torch.distributed.group") as g:
        g.WORLD = object()
        vae2.enable_window_parallel()
        with pytest.raises(RuntimeError):
            vae2.enable_tile_parallel()


def test_tiling_needs_a_row_per_rank():
    from framepack_amd.vae import _Bands

    class G:
        pass
    import unittest.mock as um
    with um.patch("torch.distributed.get_world_size", return_value=4), um.patch("torch.distributed.get_rank", return_value=1):
        with pytest.raises(ValueError):
            _Bands(None, 3)
        b = _Bands(None, 10)
        assert b.sizes == [3, 3, 2, 2] and (b.lo, b.hi) == (3, 6) and b.top and b.bot
