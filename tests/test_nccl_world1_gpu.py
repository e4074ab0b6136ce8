"""The multi-GPU paths through RCCL itself (backend "nccl"), as far as ONE GPU allows: a process group of world size 1
on cuda:0.  Every other multi-rank test runs on gloo (RCCL refuses two ranks on one device), where `_all_to_all` and the
collectives of comm.py take their host-staging branch; here the exchanges are the real `all_to_all_single(async_op=True)`
on the communicator's stream with `.wait()` on the compute stream, persistent exchange buffers reused from one attention
to the next, `all_reduce`, `all_gather_into_tensor` and `batch_isend_irecv`-free neighbour exchange (a world of one has
no neighbour) - the code the driver's multi-GPU run executes, with the data movement an identity.

One spawned process (the communicator and its streams stay out of the pytest process)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    out = {}
    try:
        from framepack_amd import comm, ops, synthetic, ulysses
        from framepack_amd.model import HunyuanVideoTransformer3DModelPacked
        from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode
        assert dist.get_backend() == "nccl"
        bf, dev = torch.bfloat16, "cuda:0"
        g = torch.Generator(device=dev).manual_seed(5)

        # ---- (1) the exchange: chunks 1 and 2, overlap callback, packed send buffer, persistent buffers, twice in a row
        heads, S, n_img = 4, 1500, 1300
        d = heads * 128
        buf = torch.randn(S, 3 * d, generator=g, device=dev).to(bf)
        q, k, v = buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:]              # row-strided views, as in the model
        wq = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(bf)
        wk = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(bf)
        ang = torch.rand(n_img, 128, generator=g, device=dev) * 6.28
        rope = torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()
        ref_buf = buf.clone()
        rq, rk, rv = ref_buf[:, :d], ref_buf[:, d:2 * d], ref_buf[:, 2 * d:]
        ops.qk_rmsnorm_rope_(rq, rk, wq, wk, rope, n_img, heads)
        want = ops.attention(rq, rk, rv, heads)

        def attn_fn(jq, jk, jv, h, o):
            return ops.attention(jq, jk, jv, h, out=o)
        bufs = {}
        side = torch.zeros(256, 512, dtype=bf, device=dev)
        wside = torch.randn(512, 512, generator=g, device=dev).to(bf)
        calls = []

        def overlap():
            calls.append(1)
            ops.gemm(side, wside, None, out=side)                          # any kernel of the caller's, under the exchange
        same = []
        for chunks in (1, 2, 2, 1):
            send = ulysses.exchange_buffers(bufs, chunks, 1, S, d // chunks, bf, torch.device(dev))["send"]
            packed = ops.qk_rmsnorm_rope_pack(q, k, v, wq, wk, rope, n_img, heads, 1, chunks, out=send)
            o = torch.full((S, d), float("nan"), dtype=bf, device=dev)
            got = ulysses.ulysses_attention(q, k, v, S, heads, 128, attn_fn, None, out=o, chunks=chunks, packed=packed,
                                            overlap=overlap, bufs=bufs)
            same.append(torch.equal(got, want))
        out["exchange_packed_bit_identical"] = all(same) and len(calls) == 4
        # without a packed buffer (the exchange packs by itself), output into a row-strided view, no overlap
        hold = torch.zeros(S, 2 * d, dtype=bf, device=dev)
        got = ulysses.ulysses_attention(rq, rk, rv, S, heads, 128, attn_fn, None, out=hold[:, d:], chunks=2)
        out["exchange_unpacked_bit_identical"] = torch.equal(hold[:, d:], want) and bool((hold[:, :d] == 0).all())
        out["pick_chunks_uses_device_cus"] = ulysses._cu_count() == torch.cuda.get_device_properties(0).multi_processor_count

        # ---- (2) comm.py through RCCL
        t = torch.arange(12, dtype=torch.float64, device=dev)
        out["all_reduce"] = torch.equal(comm.all_reduce_sum(t.clone()), t)
        x = torch.randn(5, 3, generator=g, device=dev)
        parts = comm.all_gather_padded(x, [5])
        out["all_gather_padded"] = len(parts) == 1 and torch.equal(parts[0], x)
        out["neighbour_exchange"] = comm.neighbour_exchange(None, None) == (None, None)
        out["gather_rows"] = torch.equal(ulysses.gather_rows(x), x)

        # ---- (3) the model's sequence-parallel path on the communicator (forced at world 1), TeaCache's all-reduce too
        m = HunyuanVideoTransformer3DModelPacked(**dict(synthetic.CONFIG_TINY, num_attention_heads=4)).to(bf)
        synthetic.init_synthetic_(dict(m.state_dict()), seed=0)
        m = m.to(dev).eval()
        kw = synthetic.make_inputs(dtype=bf, seed=31337, frames_latent=3, height=256, width=256, n_text=37, sigma=0.9375)
        kw = {k_: (v_.to(dev) if torch.is_tensor(v_) and v_.dtype in (bf, torch.bool) else v_) for k_, v_ in kw.items()}
        sigmas = [1.0, 0.97, 0.9, 0.5]

        def run():
            outs = []
            m.initialize_teacache(enable_teacache=True, num_steps=len(sigmas), rel_l1_thresh=0.15)
            for s in sigmas:
                kw["timestep"] = torch.tensor([s * 1000.0]).to(bf).to(dev)
                outs.append(m(return_dict=False, **kw)[0].float().cpu())
            return outs, (m.cache_hits, m.cache_misses)
        single, c1 = run()
        m.enable_sequence_parallel(force=True)
        assert m.sp_on and m.sp_world == 1
        errs = []
        for chunks in (1, 2):
            m.sp_chunks = chunks
            sharded, c2 = run()
            errs.append(max(_rel(a, b) for a, b in zip(sharded, single)))
            assert c1 == c2
        out["model_sp_err"] = max(errs)

        # ---- (4) the tiled VAE decode on the communicator (one band = the whole frame; GroupNorm moments all-reduced,
        #      k | v all-gathered, output bands gathered), default widths
        from oracle import vae_oracle as VO
        cfg_kw = dict(block_out_channels=(128, 256, 512, 512), norm_num_groups=32)
        cfg = VO.VAEConfig(**cfg_kw)
        P16 = {k_: v_.to(torch.float16) for k_, v_ in VO.random_state_dict(cfg, seed=5).items()}
        vae = AutoencoderKLHunyuanVideo(**cfg_kw)
        vae.load_state_dict(P16, strict=True)
        vae = vae.to(dev, torch.float16).eval()
        z = torch.randn(1, 16, 2, 4, 6, generator=torch.Generator().manual_seed(11))
        one = vae_decode(z.to(dev), vae).float().cpu()
        vae.enable_tile_parallel()
        tiled = vae_decode(z.to(dev), vae).float().cpu()
        out["vae_tile_max_diff"] = (tiled - one).abs().max().item()
        torch.cuda.synchronize()
        ret[0] = out
    finally:
        dist.destroy_process_group()


def test_multi_gpu_paths_run_on_rccl_with_one_rank():
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(_free_port(), ret), nprocs=1, join=True)
    r = ret[0]
    for key in ("exchange_packed_bit_identical", "exchange_unpacked_bit_identical", "pick_chunks_uses_device_cus",
                "all_reduce", "all_gather_padded", "neighbour_exchange", "gather_rows"):
        assert r[key], (key, r)
    assert r["model_sp_err"] < 4e-3, r
    assert r["vae_tile_max_diff"] <= 1e-2, r
