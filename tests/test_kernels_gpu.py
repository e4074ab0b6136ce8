"""Per-kernel parity: each HIP kernel (through the C ABI, via framepack_amd.ops) against the CPU oracle on
the same seeded inputs.  Tolerances: outputs are bf16 (8 significand bits, rel. rounding 2^-9 = 2e-3), so
elementwise kernels are held to rel-L2 <= 3e-3 and MFMA kernels (bf16 P / fp32 accumulate) to rel-L2 <= 5e-3
against an fp64/fp32 reference computed from the SAME bf16 inputs; the north-star bound is 1e-2 per block."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from framepack_amd import ops  # noqa: E402
from oracle import dit_oracle as O  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


# ------------------------------------------------------------------ K3
@pytest.mark.parametrize("rows,cols", [(1, 256), (5, 256), (777, 3072), (130, 4096), (3, 8), (19385, 3072)])
def test_layernorm_modulate(rows, cols):
    x, sc, sh = randn(rows, cols, seed=1, scale=2.0) + 0.5, randn(cols, seed=2, scale=0.3), randn(cols, seed=3)
    ref = O.layer_norm(x.float()) * (1 + sc.float()) + sh.float()
    got = ops.layernorm_modulate(x.to(DEV), sc.to(DEV), sh.to(DEV))
    assert got.dtype == BF and rel_l2(got, ref) < 3e-3
    torch.testing.assert_close(got.float().cpu(), ref, rtol=2e-2, atol=2e-2)


def test_layernorm_modulate_strided_views():
    big = randn(40, 1024, seed=4)
    x = big[:, 256:512]
    sc, sh = randn(256, seed=5), randn(256, seed=6)
    outbuf = torch.zeros(40, 768, dtype=BF, device=DEV)
    ops.layernorm_modulate(big.to(DEV)[:, 256:512], sc.to(DEV), sh.to(DEV), out=outbuf[:, 512:768])
    ref = O.layer_norm(x.float()) * (1 + sc.float()) + sh.float()
    assert rel_l2(outbuf[:, 512:768], ref) < 3e-3
    assert outbuf[:, :512].abs().max().item() == 0


# ------------------------------------------------------------------ K4
@pytest.mark.parametrize("rows,cols", [(1, 8), (333, 256), (2000, 3072)])
def test_gate_residual(rows, cols):
    x, y, g = randn(rows, cols, seed=1), randn(rows, cols, seed=2), randn(cols, seed=3)
    ref = x.float() + g.float() * y.float()
    xd = x.to(DEV)
    ops.gate_residual_(xd, y.to(DEV), g.to(DEV))
    assert rel_l2(xd, ref) < 3e-3


# ------------------------------------------------------------------ K2
@pytest.mark.parametrize("rows,rope_rows,heads,two_w", [(50, 50, 2, False), (300, 211, 3, True), (64, 0, 24, False),
                                                        (1408 + 766, 1408, 2, True)])
def test_qk_rmsnorm_rope(rows, rope_rows, heads, two_w):
    d = heads * 128
    buf = randn(rows, 3 * d + 64, seed=7)                      # q | k | v | pad  (row-strided views)
    wq, wk = 1 + 0.1 * randn(128, seed=8), 1 + 0.1 * randn(128, seed=9)
    wq2, wk2 = (1 + 0.1 * randn(128, seed=10), 1 + 0.1 * randn(128, seed=11)) if two_w else (None, None)
    g = torch.Generator("cpu").manual_seed(12)
    ang = torch.rand(max(rope_rows, 1), 128, generator=g) * 6.28
    rope = torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()

    def ref_one(x, w, w2):
        x = x.reshape(1, rows, heads, 128)
        img = O.apply_rope(O.rms_norm(x[:, :rope_rows], w), rope[:rope_rows].unsqueeze(0))
        txt = O.rms_norm(x[:, rope_rows:], w2 if w2 is not None else w)
        return torch.cat([img, txt], dim=1).reshape(rows, d)

    q_ref = ref_one(buf[:, :d], wq, wq2)
    k_ref = ref_one(buf[:, d:2 * d], wk, wk2)
    bd = buf.to(DEV)
    ops.qk_rmsnorm_rope_(bd[:, :d], bd[:, d:2 * d], wq.to(DEV), wk.to(DEV), rope.to(DEV), rope_rows, heads,
                         wq_txt=None if wq2 is None else wq2.to(DEV), wk_txt=None if wk2 is None else wk2.to(DEV))
    assert rel_l2(bd[:, :d], q_ref) < 2e-3 and rel_l2(bd[:, d:2 * d], k_ref) < 2e-3
    # same rounding points as the reference: at most 1 bf16 ulp away, and almost everywhere identical
    diff = (bd[:, :d].float().cpu() - q_ref.float()).abs()
    assert (diff <= q_ref.float().abs() * 2 ** -7 + 1e-6).all()
    assert (diff == 0).float().mean().item() > 0.97
    assert torch.equal(bd[:, 2 * d:].cpu(), buf[:, 2 * d:])     # v and pad untouched


@pytest.mark.parametrize("rows,rope_rows,heads,world,chunks", [(37, 20, 4, 2, 2), (300, 300, 24, 4, 2), (65, 0, 24, 8, 1),
                                                              (10, 7, 6, 1, 3)])
def test_qk_rmsnorm_rope_into_the_exchange_send_buffer(rows, rope_rows, heads, world, chunks):
    """The sequence-parallel form of K2: q, k, v stay as they are, and the send buffer of the head exchange holds
    exactly what the in-place kernel computes (and v), laid out [chunk][peer][row][q|k|v][heads per chunk]."""
    d = heads * 128
    buf = randn(rows, 3 * d + 64, seed=17)
    wq, wk = 1 + 0.1 * randn(128, seed=8), 1 + 0.1 * randn(128, seed=9)
    wq2, wk2 = 1 + 0.1 * randn(128, seed=10), 1 + 0.1 * randn(128, seed=11)
    g = torch.Generator("cpu").manual_seed(12)
    ang = torch.rand(max(rope_rows, 1), 128, generator=g) * 6.28
    rope = torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous().to(DEV)
    w = [x.to(DEV) for x in (wq, wk, wq2, wk2)]
    raw = buf.to(DEV)
    packed = ops.qk_rmsnorm_rope_pack(raw[:, :d], raw[:, d:2 * d], raw[:, 2 * d:3 * d], w[0], w[1], rope, rope_rows, heads,
                                      world, chunks, wq_txt=w[2], wk_txt=w[3])
    assert torch.equal(raw.cpu(), buf)                                   # inputs untouched
    done = buf.to(DEV)
    ops.qk_rmsnorm_rope_(done[:, :d], done[:, d:2 * d], w[0], w[1], rope, rope_rows, heads, wq_txt=w[2], wk_txt=w[3])
    hwc = heads // world // chunks * 128
    want = torch.stack([done[:, i * d:(i + 1) * d].reshape(rows, world, chunks, hwc) for i in range(3)], dim=3) \
        .permute(2, 1, 0, 3, 4).reshape(chunks, world, rows, 3 * hwc)
    assert tuple(packed.shape) == (chunks, world, rows, 3 * hwc) and torch.equal(packed, want)


# ------------------------------------------------------------------ K7 / K10
@pytest.mark.parametrize("n_in,n_out,silu", [(256, 1536, True), (3072, 18432, True), (3072, 100, False), (4096, 7, True)])
def test_silu_gemv(n_in, n_out, silu):
    t, W, b = randn(n_in, seed=1), randn(n_out, n_in, seed=2, scale=1 / math.sqrt(n_in)), randn(n_out, seed=3)
    a = torch.nn.functional.silu(t) if silu else t              # bf16 SiLU, as the reference
    ref = a.float() @ W.float().T + b.float()
    got = ops.silu_gemv(t.to(DEV), W.to(DEV), b.to(DEV), apply_silu=silu)
    assert rel_l2(got, ref) < 4e-3


def test_rel_l1():
    a, b = randn(1000, 3072, seed=1), randn(1000, 3072, seed=2)
    acc = ops.rel_l1(a.to(DEV), b.to(DEV)).cpu()
    d = (a - b).abs().float().sum().item()                      # bf16 subtraction, as the reference
    m = b.float().abs().sum().item()
    assert abs(acc[0].item() - d) / d < 1e-4 and abs(acc[1].item() - m) / m < 1e-4


# ------------------------------------------------------------------ K5
@pytest.fixture(params=[0, 1], ids=["eight_wave", "four_wave"])
def gemm_variant(request):
    """Both main loop generations the library holds (fp_debug_gemm_variant), the default restored afterwards."""
    from framepack_amd import lib as fplib
    old = fplib.load().fp_debug_gemm_variant(request.param)
    yield request.param
    fplib.load().fp_debug_gemm_variant(old)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1, 8, 64), (300, 768, 256), (1408, 1792, 256), (777, 3072, 3072),
                                   (513, 264, 1280)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(M, N, K, epi, gemm_variant):
    a, w, b = randn(M, K, seed=1), randn(N, K, seed=2, scale=1 / math.sqrt(K)), randn(N, seed=3)
    lin = a.float() @ w.float().T + b.float()
    gelu = torch.nn.functional.gelu(lin, approximate="tanh")
    kw = {}
    if epi == 0:
        ref = lin
    elif epi == 1:
        ref = gelu
    elif epi == 2:
        gf = (N // 2) // 4 * 4
        ref = torch.cat([lin[:, :gf], gelu[:, gf:]], dim=1)
        kw = dict(gelu_from=gf)
    else:
        r, g = randn(M, N, seed=4), randn(N, seed=5)
        ref = r.float() + g.float() * lin
        kw = dict(resid=r.to(DEV), gate=g.to(DEV))
    got = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), epilogue=epi, **kw)
    assert got.shape == (M, N) and rel_l2(got, ref) < 4e-3


def test_gemm_strided_two_segment_and_inplace_residual(gemm_variant):
    M, N, K1, K2 = 389, 512, 256, 1024
    buf = randn(M, 3 * 256 + K2 + 128, seed=1)                  # [k | v | o | mlp | pad]
    w, b = randn(N, K1 + K2, seed=2, scale=0.03), randn(N, seed=3)
    x, g = randn(M, N, seed=4), randn(N, seed=5)
    bd = buf.to(DEV)
    a_cat = torch.cat([buf[:, 512:768], buf[:, 768:768 + K2]], dim=1)
    ref = x.float() + g.float() * (a_cat.float() @ w.float().T + b.float())
    # (a) contiguous column range of a row-strided buffer
    xd = x.to(DEV)
    ops.gemm(bd[:, 512:768 + K2], w.to(DEV), b.to(DEV), out=xd, epilogue=3, resid=xd, gate=g.to(DEV))
    assert rel_l2(xd, ref) < 4e-3
    # (b) two explicit K segments
    xd2 = x.to(DEV)
    ops.gemm(bd[:, 512:768], w.to(DEV), b.to(DEV), out=xd2, epilogue=3, resid=xd2, gate=g.to(DEV),
             a2=bd[:, 768:768 + K2])
    assert rel_l2(xd2, ref) < 4e-3


@pytest.mark.parametrize("N,K,epi", [(21504, 3072, 2), (3072, 15360, 3), (9216, 3072, 0), (12288, 3072, 1)])
def test_gemm_full_size_sampled_rows(N, K, epi):
    """The model's GEMM shapes at the BASELINE config 2 row count (S = 19385: 76 row tiles, the last one ragged, more
    tiles than CUs so the persistent workgroups loop): sampled rows of every region of the output against fp64, plus
    a size-independent property - doubling A doubles the linear part exactly (powers of two commute with rounding)."""
    M = 19385
    g = torch.Generator(device=DEV).manual_seed(7)
    a = torch.randn(M, K, generator=g, device=DEV).to(BF)
    w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(BF)
    b = torch.randn(N, generator=g, device=DEV).to(BF)
    kw, gf = {}, 0
    if epi == 2:
        gf = 9216
        kw = dict(gelu_from=gf)
    if epi == 3:
        r, gate = torch.randn(M, N, generator=g, device=DEV).to(BF), torch.randn(N, generator=g, device=DEV).to(BF)
        kw = dict(resid=r, gate=gate)
    got = ops.gemm(a, w, b, epilogue=epi, **kw)
    rows = torch.tensor([0, 1, 127, 128, 255, 256, 4095, 9999, 19199, 19200, 19328, 19383, 19384], device=DEV)
    lin = (a[rows].double() @ w.double().T + b.double()).cpu()
    gelu = torch.nn.functional.gelu(lin, approximate="tanh")
    if epi == 0:
        ref = lin
    elif epi == 1:
        ref = gelu
    elif epi == 2:
        ref = torch.cat([lin[:, :gf], gelu[:, gf:]], dim=1)
    else:
        ref = r[rows].double().cpu() + gate.double().cpu() * lin
    assert rel_l2(got[rows], ref) < 4e-3
    torch.testing.assert_close(got[rows].double().cpu(), ref, rtol=2e-2, atol=2e-2)
    if epi == 0:
        got2 = ops.gemm(a * 2, w, None)
        got1 = ops.gemm(a, w, None)
        assert torch.equal(got2, got1 * 2)


def test_gemm_rejects_bad_arguments():
    a, w = randn(4, 100, seed=1).to(DEV), randn(8, 100, seed=2).to(DEV)   # K % 64 != 0
    with pytest.raises(ValueError):
        ops.gemm(a, w, None)
    with pytest.raises(RuntimeError):
        ops.gemm(randn(4, 64), randn(8, 64), None)                        # CPU tensors: no fallback


# ------------------------------------------------------------------ K1
@pytest.fixture(params=[1, 2], ids=["gen2_agpr_scores", "gen3_vgpr_scores"])
def attn_variant(request):
    """Both attention sweeps the library holds (fp_debug_attn_variant), the default restored afterwards."""
    from framepack_amd import lib as fplib
    old = fplib.load().fp_debug_attn_variant(request.param)
    yield request.param
    fplib.load().fp_debug_attn_variant(old)


def attn_ref(q, k, v, heads):
    S = q.shape[0]
    return O.attention_reference_f64(q.view(S, heads, 128), k.view(S, heads, 128), v.view(S, heads, 128)).reshape(S, -1)


@pytest.mark.parametrize("S,heads", [(1, 1), (63, 2), (64, 1), (65, 3), (255, 1), (256, 2), (257, 1), (777, 3),
                                     (1408 + 766, 2), (2 * 256 + 64, 24)])
def test_attention_single_segment(S, heads, attn_variant):
    d = heads * 128
    q, k, v = randn(S, d, seed=1), randn(S, d, seed=2), randn(S, d, seed=3)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads)
    assert rel_l2(got, attn_ref(q, k, v, heads)) < 5e-3


def test_attention_strided_fused_buffer_and_alias(attn_variant):
    """q,k,v as column ranges of one [S, k|v|q|mlp] buffer (the single-stream layout); o written over q."""
    S, heads = 900, 2
    d = heads * 128
    buf = randn(S, 3 * d + 512, seed=5)
    ref = attn_ref(buf[:, 2 * d:3 * d].contiguous(), buf[:, :d].contiguous(), buf[:, d:2 * d].contiguous(), heads)
    bd = buf.to(DEV)
    ops.attention(bd[:, 2 * d:3 * d], bd[:, :d], bd[:, d:2 * d], heads, out=bd[:, 2 * d:3 * d])
    assert rel_l2(bd[:, 2 * d:3 * d], ref) < 5e-3
    assert torch.equal(bd[:, :2 * d].cpu(), buf[:, :2 * d]) and torch.equal(bd[:, 3 * d:].cpu(), buf[:, 3 * d:])


def test_attention_varlen_segments(attn_variant):
    """cu_seqlens as ref hunyuan_video_packed.py:63-77 builds them: valid tokens and the padding tail of
    each sample are separate segments; an empty segment is allowed."""
    heads, d = 2, 256
    cu = [0, 300, 517, 517, 1000, 1064]
    S = cu[-1]
    q, k, v = randn(S, d, seed=1), randn(S, d, seed=2), randn(S, d, seed=3)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads,
                        cu_seqlens=torch.tensor(cu, dtype=torch.int32, device=DEV), max_seqlen=483)
    for a, b in zip(cu[:-1], cu[1:]):
        if b > a:
            assert rel_l2(got[a:b], attn_ref(q[a:b], k[a:b], v[a:b], heads)) < 5e-3


def test_attention_max_jump_forces_rescale_branch(attn_variant):
    """Spike one key against every query in a late tile so the running max jumps there
    (the data-dependent rescale branch), plus large-magnitude scores."""
    S, heads = 640, 1
    q, k, v = randn(S, 128, seed=1), randn(S, 128, seed=2), randn(S, 128, seed=3)
    k[500] = (q[:64].float().mean(0) * 6).to(BF)
    k[630] = (q[100:164].float().mean(0) * -9).to(BF)
    q[7] = q[7] * 8
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads)
    ref = attn_ref(q, k, v, heads)
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < 5e-3
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=3e-2, atol=3e-2)


def test_attention_growth_beyond_the_deferred_max_threshold(attn_variant):
    """The third-generation sweep keeps the first tile's row max as its reference and only moves it when a row's max
    grows by more than 2^60 (its slow path: true max, O / l rescaled, the tile and the next tile's accumulators
    shifted, P recomputed).  All queries lie near one direction b; key 100 = 5.9 b and key 500 = 6 b score ~96 and ~98
    in the log2 domain against a first-tile max near 2: tile 1 must take the slow path, tile 7 then runs with a
    reference that is stale by ~1.6, and both spikes carry weight in the result (0.24 : 0.76)."""
    S, heads = 777, 2
    g = torch.Generator("cpu").manual_seed(21)
    b = torch.randn(heads, 128, generator=g)
    b = b / b.norm(dim=-1, keepdim=True) * math.sqrt(128.0)
    q = (b.unsqueeze(0) + 0.1 * torch.randn(S, heads, 128, generator=g)).reshape(S, -1).to(BF)
    k = torch.randn(S, heads, 128, generator=g)
    k[100], k[500] = 5.9 * b, 6.0 * b
    k = k.reshape(S, -1).to(BF)
    v = randn(S, heads * 128, seed=3)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads)
    ref = attn_ref(q, k, v, heads)
    assert torch.isfinite(got.float()).all()
    w = (ref.float() - v[500].float()).norm() / v[500].float().norm()
    assert w > 0.1                                   # not a one-hot softmax: key 100 matters too
    assert rel_l2(got, ref) < 5e-3
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=3e-2, atol=3e-2)


def test_attention_full_size_properties():
    """BASELINE config 2 shape (S = 19385, 24 heads): size-independent properties + sampled rows vs fp64."""
    S, heads = 19385, 24
    d = heads * 128
    g = torch.Generator(device=DEV).manual_seed(0)
    q = torch.randn(S, d, generator=g, device=DEV).to(BF)
    k = torch.randn(S, d, generator=g, device=DEV).to(BF)
    v = torch.randn(S, d, generator=g, device=DEV).to(BF)
    o = ops.attention(q, k, v, heads)
    # (1) V = const -> O = const exactly up to bf16 rounding (softmax weights sum to 1; tail masking right)
    o1 = ops.attention(q, k, torch.full_like(v, 0.75), heads)
    assert (o1.float() - 0.75).abs().max().item() < 4e-3
    # (2) permuting the keys/values leaves the output unchanged up to accumulation order
    perm = torch.randperm(S, generator=g, device=DEV)
    o2 = ops.attention(q, k[perm].contiguous(), v[perm].contiguous(), heads)
    assert rel_l2(o2, o) < 4e-3
    # (3) sampled query rows x 3 heads against the fp64 oracle
    rows = torch.tensor([0, 1, 255, 256, 9000, 18399, 18400, 19200, 19384])
    for h in (0, 11, 23):
        sl = slice(h * 128, (h + 1) * 128)
        qh, kh, vh = q[rows][:, sl].cpu().double(), k[:, sl].cpu().double(), v[:, sl].cpu().double()
        p = torch.softmax(qh @ kh.T / math.sqrt(128.0), dim=-1)
        assert rel_l2(o[rows][:, sl], p @ vh) < 6e-3


def test_attention_split_last_round_matches_unsplit_sweeps(attn_variant):
    """261 work items on a 256-CU chip: the 5 items of the last round are swept as 8 KV ranges each and merged
    (fp_attn_fwd_bf16 with a workspace).  Their rows, and rows of unsplit items, against an fp32 reference; the
    ragged last query block and last KV tile are inside the split items."""
    S, heads = 86 * 256 + 10, 3
    g = torch.Generator(device=DEV).manual_seed(5)
    q, k, v = (torch.randn(S, heads * 128, generator=g, device=DEV).to(BF) for _ in range(3))
    o = ops.attention(q, k, v, heads)
    assert torch.isfinite(o.float()).all()
    for h, lo, hi in ((2, 82 * 256, S), (2, 0, 300), (0, 21000, S), (1, 11000, 11300)):
        sl = slice(h * 128, (h + 1) * 128)
        p = torch.softmax(q[lo:hi, sl].float() @ k[:, sl].float().T / math.sqrt(128.0), dim=-1)
        assert rel_l2(o[lo:hi, sl], p @ v[:, sl].float()) < 5e-3, (h, lo)


@pytest.mark.parametrize("heads", [6, 12])
def test_attention_splits_a_nearly_full_last_round(heads):
    """The product sequence (76 query blocks) with the 6 / 12 heads one rank of a 4- / 2-way sequence-parallel group
    attends: 456 / 912 items = 1 / 3 full rounds + 200 / 144 items on 256 CUs.  With two partial slots per CU in the
    workspace the 144 are cut 3 ways (an unsplit partial round would idle 112 CUs for a whole sweep; measured -4.4 %);
    the 200 could only be cut in two, which would take a sweep again, so there the last 32 items are cut 8 ways.
    Against the unsplit kernel (no workspace) on every row, and sampled rows against fp32 math."""
    from framepack_amd import lib as fplib
    S = 19385
    g = torch.Generator(device=DEV).manual_seed(heads)
    q, k, v = (torch.randn(S, heads * 128, generator=g, device=DEV).to(BF) for _ in range(3))
    L = fplib.load()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    assert (L.fp_attn_workspace_bytes() - 256) // ((256 * 128 + 512) * 4) == 2 * ncu
    o = ops.attention(q, k, v, heads)
    o0 = torch.full_like(o, float("nan"))
    fplib.check(L.fp_attn_fwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o0.data_ptr(), q.stride(0), k.stride(0),
                                   v.stride(0), o0.stride(0), None, 1, S, S, heads, 128, 1.0 / math.sqrt(128.0), None, 0,
                                   torch.cuda.current_stream().cuda_stream), "fp_attn_fwd_bf16")
    assert torch.isfinite(o.float()).all() and torch.isfinite(o0.float()).all()
    assert rel_l2(o, o0) < 2e-3
    assert (o.float() - o0.float()).abs().max().item() < 2e-2
    for h, lo, hi in ((heads - 1, 75 * 256, S), (heads - 1, 40 * 256, 40 * 256 + 200), (0, 0, 256)):
        sl = slice(h * 128, (h + 1) * 128)
        p = torch.softmax(q[lo:hi, sl].float() @ k[:, sl].float().T / math.sqrt(128.0), dim=-1)
        assert rel_l2(o[lo:hi, sl], p @ v[:, sl].float()) < 5e-3, (h, lo)


@pytest.mark.parametrize("layout", ["double_img", "double_txt", "single"])
@pytest.mark.parametrize("M", [1, 300, 777])
def test_gemm_with_k2_in_its_epilogue_is_bit_identical_to_the_two_passes(layout, M):
    """fp_gemm_bf16_qknorm: the q | k | v projection (or the single block's in-projection with its GELU columns) with the
    per-head RMSNorm and RoPE applied in the epilogue's row layout - the same arithmetic on the same bf16-rounded values
    as fp_gemm_bf16 followed by fp_qk_rmsnorm_rope_bf16, so the outputs must be the same BITS."""
    heads, K = 4, 256
    d = heads * 128
    g = torch.Generator().manual_seed(M + len(layout))
    if layout == "single":
        N, qcol0, kcol0, epi, gelu_from = 7 * d, 2 * d, 0, ops.EPI_GELU_FROM, 3 * d        # [k | v | q | mlp]
        rope_rows = (2 * M) // 3
        wq_txt = wk_txt = None
    else:
        N, qcol0, kcol0, epi, gelu_from = 3 * d, 0, d, ops.EPI_BIAS, 0                       # [q | k | v]
        rope_rows = M if layout == "double_img" else 0
        wq_txt, wk_txt = ((torch.randn(128, generator=g) * 0.2 + 1).to(BF).to(DEV) for _ in range(2))
        if layout == "double_img":
            wq_txt = wk_txt = None
    a = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = (torch.randn(N, K, generator=g) / 16).to(BF).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(BF).to(DEV)
    wq, wk = ((torch.randn(128, generator=g) * 0.2 + 1).to(BF).to(DEV) for _ in range(2))
    ang = torch.rand(max(rope_rows, 1), 128, generator=g) * 6.28
    rope = torch.cat([torch.cos(ang), torch.sin(ang)], dim=1).contiguous().to(DEV)
    want = ops.gemm(a, w, b, epilogue=epi, gelu_from=gelu_from)
    ops.qk_rmsnorm_rope_(want[:, qcol0:qcol0 + d], want[:, kcol0:kcol0 + d], wq, wk, rope, rope_rows, heads,
                         wq_txt=wq_txt, wk_txt=wk_txt)
    got = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    ops.gemm_qknorm(a, w, b, got, wq, wk, rope, rope_rows, qcol0, kcol0, heads, epilogue=epi, gelu_from=gelu_from,
                    wq_txt=wq_txt, wk_txt=wk_txt)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()


@pytest.mark.parametrize("M,m_split,N,K", [(19385, 18400, 3072, 3072), (1500, 1100, 768, 256), (700, 300, 9216, 128),
                                          (520, 519, 512, 64), (300, 0, 512, 64), (300, 300, 512, 64),
                                          (600, 200, 12288, 64), (1000, 300, 8192, 128), (513, 1, 16384, 64)])
@pytest.mark.parametrize("epi", ["bias", "gelu", "gate_residual"])
def test_gemm_over_two_row_segments_is_bit_identical_to_two_gemms(M, m_split, N, K, epi):
    """fp_gemm_bf16_rows2: rows [0, m_split) with (w0, bias0, gate0), rows [m_split, M) with (w1, bias1, gate1), one
    launch (ref hunyuan_video_packed.py:537-604: the image and the text linear of a double-stream block).  Row-strided a /
    out views (the model's q over Y), in-place residual.  Ragged tiles on both sides of the split, empty segments."""
    g = torch.Generator(device=DEV).manual_seed(M + N)
    abuf = torch.randn(M, K + 64, generator=g, device=DEV).to(BF)
    a = abuf[:, :K]
    w0, w1 = ((torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(BF) for _ in range(2))
    b0, b1 = ((torch.randn(N, generator=g, device=DEV) * 0.1).to(BF) for _ in range(2))
    g0, g1 = (torch.randn(N, generator=g, device=DEV).to(BF) for _ in range(2))
    x = torch.randn(M, N + 8, generator=g, device=DEV).to(BF)
    e = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_GELU, "gate_residual": ops.EPI_GATE_RESIDUAL}[epi]
    kw0 = dict(epilogue=e) if epi != "gate_residual" else dict(epilogue=e, gate=g0)
    kw1 = dict(epilogue=e) if epi != "gate_residual" else dict(epilogue=e, gate=g1)
    want = x.clone()
    wv = want[:, :N]
    if epi == "gate_residual":
        kw0["resid"], kw1["resid"] = wv[:m_split], wv[m_split:]
    ops.gemm(a[:m_split], w0, b0, out=wv[:m_split], **kw0)
    ops.gemm(a[m_split:], w1, b1, out=wv[m_split:], **kw1)
    got = x.clone()
    gv = got[:, :N]
    ops.gemm_rows2(a, m_split, w0, w1, b0, b1, out=gv, epilogue=e, resid=gv if epi == "gate_residual" else None,
                   gate0=g0 if epi == "gate_residual" else None, gate1=g1 if epi == "gate_residual" else None)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want)
    if epi == "bias" and M <= 1500:
        ref = torch.cat([a[:m_split].float() @ w0.float().T + b0.float(), a[m_split:].float() @ w1.float().T + b1.float()])
        assert rel_l2(gv, ref) < 4e-3


def test_layernorm_affine_matches_torch_layer_norm():
    """K3's affine form (the token refiner's nn.LayerNorm, ref hunyuan_video_packed.py:264, 273 through
    dit_common.py:9-10): fp32 arithmetic, one rounding - against F.layer_norm in fp32 on the same bf16 values."""
    g = torch.Generator(device=DEV).manual_seed(21)
    for rows, cols in ((512, 3072), (37, 256), (5, 4096)):
        big = torch.randn(rows, cols + 64, generator=g, device=DEV).to(BF)
        x = big[:, :cols]                                                   # row-strided view
        w = (1 + 0.3 * torch.randn(cols, generator=g, device=DEV)).to(BF)
        w[:3] = torch.tensor([0.0101, -2.5, 5.3], device=DEV).to(BF)       # weights far from 1 (w - 1 is not a bf16 value)
        b = torch.randn(cols, generator=g, device=DEV).to(BF)
        ref = torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-6)
        got = ops.layernorm_affine(x, w, b, 1e-6)
        assert got.dtype == BF and tuple(got.shape) == (rows, cols)
        assert rel_l2(got, ref) < 3e-3
        assert (got.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7


def test_gemm_bf16_with_fp32_output_is_the_fp32_linear():
    """The fp32 head (ref :1048-1053: activations and weights upcast to fp32, then nn.Linear): bf16 operands hold the same
    values as their fp32 copies and bf16 x bf16 products are exact in fp32, so K5 with fp32 output must agree with the fp32
    linear to summation order (1e-6 relative), far inside what a bf16-rounded output would give (4e-3)."""
    g = torch.Generator(device=DEV).manual_seed(22)
    for M, N, K in ((14400, 64, 3072), (300, 64, 256), (1000, 512, 512)):
        a = torch.randn(M, K, generator=g, device=DEV).to(BF)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(BF)
        b = torch.randn(N, generator=g, device=DEV)
        got = ops.gemm_f32out(a, w, b)
        ref = a.double() @ w.double().T + b.double()
        assert got.dtype == torch.float32 and rel_l2(got, ref) < 2e-6
        assert rel_l2(ops.gemm_f32out(a, w, None), ref - b.double()) < 2e-6
