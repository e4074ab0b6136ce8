"""The sequence-parallel forward of the model (image AND text rows sharded, Ulysses exchange around every attention)
against the same model's single-GPU forward: two processes on ONE GPU over a gloo group (RCCL does not accept two
ranks on one device; the exchange then goes through host memory, everything else is the production path)."""
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


def _worker(rank, world, port, case, teacache, fp8_attention, heads, chunks, ret, wide=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from framepack_amd import synthetic
        from framepack_amd.model import HunyuanVideoTransformer3DModelPacked
        bf, dev = torch.bfloat16, "cuda:0"
        if wide:        # 13B width (24 heads x 128, mlp 12 288), one double + one single block; `heads` other than 24
            # keeps the head width and shrinks the model's (12 heads on 4 ranks = the 3 heads a rank of 8 holds)
            m = HunyuanVideoTransformer3DModelPacked(**dict(synthetic.CONFIG_13B, num_layers=1, num_single_layers=1,
                                                            num_attention_heads=heads)).to(bf)
        else:
            m = HunyuanVideoTransformer3DModelPacked(**dict(synthetic.CONFIG_TINY, num_attention_heads=heads)).to(bf)
        synthetic.init_synthetic_(dict(m.state_dict()), seed=0)
        m = m.to(dev).eval()
        if wide:
            from framepack_amd.ulysses import pick_chunks
            assert m.heads == heads and pick_chunks(19385, heads // world) == (1 if (heads // world) % 2 else 2)
        if fp8_attention:
            m.enable_fp8_attention()       # per-head scales over all rows: the same on one GPU and on a head group
        kw = synthetic.make_inputs(dtype=bf, **case)
        kw = {k: (v.to(dev) if torch.is_tensor(v) and v.dtype in (bf, torch.bool) else v) for k, v in kw.items()}
        sigmas = [1.0, 0.97, 0.9, 0.5] if teacache else [kw["timestep"].item() / 1000.0]

        def run():
            outs = []
            if teacache:
                m.initialize_teacache(enable_teacache=True, num_steps=len(sigmas), rel_l1_thresh=0.15)
            for s in sigmas:
                kw["timestep"] = torch.tensor([s * 1000.0]).to(bf).to(dev)
                outs.append(m(return_dict=False, **kw)[0].float().cpu())
            return outs, (m.cache_hits, m.cache_misses) if teacache else None
        single, c1 = run()
        m.enable_sequence_parallel()
        m.sp_chunks = chunks
        sharded, c2 = run()
        err = max(((a - b).norm() / b.norm()).item() for a, b in zip(sharded, single))
        # the single blocks' in-projection as two GEMMs around the exchange (model._single_block, the default) against
        # the one fused launch: the same arithmetic per output element
        assert m.sp_overlap_mlp
        m.sp_overlap_mlp = False
        fused, c3 = run()
        same = all(torch.equal(a, b) for a, b in zip(fused, sharded))
        ret[rank] = (err, c1 == c2 and c2 == c3, same)
    finally:
        dist.destroy_process_group()


CASES = {
    "config1": dict(seed=31337, frames_latent=3, height=256, width=256, n_text=37, sigma=0.9375),
    "odd_text": dict(seed=8, frames_latent=2, height=96, width=160, n_text=1, sigma=0.05),
    # BASELINE config 3's shape: the 640 x 640 x 33-frame section with 256 text tokens, S = 19 385
    "config3": dict(seed=31337, frames_latent=9, height=640, width=640, n_text=256, sigma=0.7),
    # BASELINE config 4's stress point: latent window 33 at 640 x 640, the longest packed sequence (S = 57 785)
    "config4": dict(seed=31337, frames_latent=33, height=640, width=640, n_text=256, sigma=0.7),
}


@pytest.mark.parametrize("case,teacache,fp8_attention,heads,chunks", [
    ("config1", False, False, 2, None), ("odd_text", False, False, 2, None), ("config1", True, False, 2, None),
    ("config1", False, True, 2, None),
    ("config1", False, False, 4, 2)])        # two head chunks per rank: the overlapped exchange of ulysses.py
def test_sequence_parallel_forward_equals_single_gpu(case, teacache, fp8_attention, heads, chunks):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), CASES[case], teacache, fp8_attention, heads, chunks, ret), nprocs=world,
             join=True)
    assert len(ret) == world
    for err, same_counts, split_is_bit_identical in ret.values():
        # same kernels on the same values; only the order of the keys inside the attention (and the tile split of
        # the GEMMs) differs between the two runs
        assert err < 4e-3 and same_counts
        assert split_is_bit_identical


@pytest.mark.parametrize("case,world,heads", [("config3", 2, 24), ("config4", 2, 24), ("config3", 4, 24), ("config3", 4, 12)],
                         ids=["config3_S19385", "config4_max_packed_S57785", "config3_four_ranks", "config3_three_heads_per_rank"])
def test_sequence_parallel_forward_at_13b_width_and_product_length(case, world, heads):
    """BASELINE configs 3 and 4 as far as one GPU goes: 13B-width blocks (1 double + 1 single) at S = 19 385 and at the
    longest packed sequence (window 33: S = 57 785), two ranks (12 heads each: `pick_chunks` = 2, the single block's
    in-projection split around the exchange) against the same model's single-GPU forward; the 4-GPU shape (6 heads per
    rank, two chunks of 3; S = 19 385 is not a multiple of 4: the last rank's rows are padded) and, on 4 ranks with a
    12-head model of the same head width, the 3 heads per rank (one exchange, `pick_chunks` = 1) that a rank of the 8-GPU
    run holds.  The scaling curve itself is unmeasured here (one GPU)."""
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), CASES[case], False, False, heads, None, ret, True), nprocs=world,
             join=True)
    assert len(ret) == world
    for err, same_counts, split_is_bit_identical in ret.values():
        assert err < 4e-3 and same_counts
        assert split_is_bit_identical
