"""K11 parity: the causal-3D VAE decode kernels (through the C ABI) and the whole decoder against the CPU oracle
(oracle/vae_oracle.py, PARITY UNPINNED: it restates diffusers' AutoencoderKLHunyuanVideo, absent offline).

Tolerances: volumes are fp16 (11 significand bits, rel. rounding 2^-11 = 5e-4); a convolution with fp32
accumulation is held to rel-L2 <= 2e-3 against an fp32 reference computed from the same fp16 inputs; the
north-star bound for decoded frames is PSNR >= 40 dB (peak-to-peak 2.0, frames live in [-1, 1])."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from framepack_amd import ops  # noqa: E402
from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode, vae_encode  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402

DEV = "cuda:0"
H16 = torch.float16


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def psnr(a, b, peak=2.0):
    mse = ((a.double().cpu() - b.double().cpu()) ** 2).mean().item()
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


def rand(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(H16)


def prep_w(w):          # [Cout, Cin, 3, 3, 3] -> [Cout, 27 * Cin] tap-major
    return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous()


def ref_conv(x_cl, w, b, up=(1, 1, 1), resid=None):
    """x_cl [T,H,W,C] fp16 -> fp32 reference in channels-last, through the oracle's upsample + causal conv."""
    x = x_cl.float().permute(3, 0, 1, 2).unsqueeze(0)
    P = {"c.conv.weight": w.float(), "c.conv.bias": b.float()}
    if up != (1, 1, 1):
        y = VO.upsample_causal({"u.conv.conv.weight": w.float(), "u.conv.conv.bias": b.float()}, "u", x, up)
    else:
        y = VO.causal_conv3d(P, "c", x)
    y = y[0].permute(1, 2, 3, 0)
    return y + resid.float() if resid is not None else y


@pytest.mark.parametrize("T,H,W,cin,cout,up,with_res", [
    (3, 8, 8, 64, 64, (1, 1, 1), False),
    (2, 5, 7, 128, 72, (1, 1, 1), True),          # ragged M (70 rows) and N tails
    (1, 16, 20, 64, 256, (1, 1, 1), False),       # single frame: all temporal taps clamp to frame 0
    (3, 6, 6, 64, 64, (1, 2, 2), False),          # spatial upsampling folded into the gather
    (3, 5, 6, 64, 128, (2, 2, 2), True),          # temporal + spatial (first frame not doubled)
    (1, 4, 4, 64, 64, (2, 2, 2), False),          # one frame in: one frame out
    (5, 40, 40, 128, 128, (1, 1, 1), True),       # several M tiles
    (5, 32, 48, 64, 128, (1, 1, 1), True),        # a frame = 6 whole tiles
    (3, 16, 16, 64, 256, (2, 2, 2), False),       # whole tiles per frame, with upsampling
    (3, 9, 11, 192, 72, (1, 1, 1), True),         # Cin not a power of two: the 8-wave kernel's coordinate gather
    (2, 6, 5, 192, 136, (2, 2, 2), False),        # the same with upsampling
])
def test_conv3d_causal(T, H, W, cin, cout, up, with_res):
    x = rand(T, H, W, cin, seed=1)
    w = rand(cout, cin, 3, 3, 3, seed=2, scale=(27 * cin) ** -0.5)
    b = rand(cout, seed=3, scale=0.1)
    To = 2 * T - 1 if up[0] == 2 else T
    res = rand(To, H * up[1], W * up[2], cout, seed=4) if with_res else None
    ref = ref_conv(x, w, b, up, res)
    got = ops.conv3d_causal(x.to(DEV), prep_w(w).to(DEV), b.to(DEV), up=up, resid=None if res is None else res.to(DEV))
    assert got.dtype == H16 and tuple(got.shape) == tuple(ref.shape)
    assert rel_l2(got, ref) < 2e-3
    torch.testing.assert_close(got.float().cpu(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("T,H,W,cin,cout,stride", [
    (1, 16, 16, 64, 64, (1, 2, 2)),               # the single-image case: every temporal tap clamps to frame 0
    (5, 12, 10, 64, 72, (2, 2, 2)),               # 5 -> 3 frames
    (4, 9, 7, 128, 64, (2, 2, 2)),                # even T (4 -> 2) and odd H, W (9 -> 5, 7 -> 4): upper clamps
    (3, 6, 6, 64, 128, (2, 1, 1)),                # temporal stride only
    (3, 40, 40, 64, 256, (1, 2, 2)),              # several M tiles
    (4, 9, 7, 192, 64, (2, 2, 2)),                # Cin not a power of two
])
def test_conv3d_causal_strided(T, H, W, cin, cout, stride):
    x = rand(T, H, W, cin, seed=1)
    w = rand(cout, cin, 3, 3, 3, seed=2, scale=(27 * cin) ** -0.5)
    b = rand(cout, seed=3, scale=0.1)
    xr = x.float().permute(3, 0, 1, 2).unsqueeze(0)
    ref = VO.causal_conv3d({"c.conv.weight": w.float(), "c.conv.bias": b.float()}, "c", xr, stride)[0].permute(1, 2, 3, 0)
    got = ops.conv3d_causal_strided(x.to(DEV), prep_w(w).to(DEV), b.to(DEV), stride)
    assert got.dtype == H16 and tuple(got.shape) == tuple(ref.shape)
    assert rel_l2(got, ref) < 2e-3
    torch.testing.assert_close(got.float().cpu(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("T,H,W,cin,with_res", [
    (2, 384, 384, 128, True),        # whole 512-row tiles, one band walk (a frame = 288 tiles)
    (3, 300, 301, 128, False),       # ragged last tile, frames that are no whole number of tiles
    (2, 384, 400, 256, True),        # the last up block's first convolution (256 -> 128)
])
def test_conv3d_tall_tile_is_bit_identical_to_the_narrow_tile(T, H, W, cin, with_res):
    """Cout = 128 over >= 2 rounds of 512-row tiles runs on the 512 x 128 tile of gemm_w4.h (TALL); with
    fp_debug_gemm_variant(1 | 4) the same call runs on the 256 x 128 tile that the cases of test_conv3d_causal pin to
    the oracle.  Same K order, same MFMA per output element: the same bits."""
    from framepack_amd import lib as fplib
    g = torch.Generator(device=DEV).manual_seed(T * H + W)
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(H16)
    w = (torch.randn(128, 27 * cin, generator=g, device=DEV) * (27 * cin) ** -0.5).to(H16)
    b = (torch.randn(128, generator=g, device=DEV) * 0.1).to(H16)
    res = torch.randn(T, H, W, 128, generator=g, device=DEV).to(H16) if with_res else None
    assert T * H * W >= 2 * 512 * torch.cuda.get_device_properties(0).multi_processor_count
    tall = ops.conv3d_causal(x, w, b, resid=res)
    old = fplib.load().fp_debug_gemm_variant(1 | 4)
    try:
        narrow = ops.conv3d_causal(x, w, b, resid=res)
    finally:
        fplib.load().fp_debug_gemm_variant(old)
    assert torch.isfinite(tall.float()).all()
    assert torch.equal(tall, narrow), (tall.float() - narrow.float()).abs().max().item()
    # and a corner of the volume against fp32 math (first frame, first rows: replicate padding and the causal clamp)
    crop = x[:, :6, :7].cpu()
    wr = w.cpu().reshape(128, 3, 3, 3, cin).permute(0, 4, 1, 2, 3)
    ref = ref_conv(crop, wr, b.cpu(), (1, 1, 1), None)
    got = tall[:, :5, :6].float().cpu() - (res[:, :5, :6].float().cpu() if with_res else 0)
    assert rel_l2(got, ref[:, :5, :6]) < 3e-3


def test_conv3d_rejects_bad_arguments():
    x, w = rand(2, 4, 4, 32).to(DEV), rand(64, 27 * 32).to(DEV)
    with pytest.raises(ValueError):
        ops.conv3d_causal(x, w, None)                              # Cin % 64 != 0
    with pytest.raises(RuntimeError):
        ops.conv3d_causal(rand(2, 4, 4, 64), rand(64, 27 * 64), None)   # CPU tensors: no fallback


@pytest.mark.parametrize("rows_shape,C,groups,silu", [((3, 8, 8), 64, 16, True), ((2, 5, 7), 128, 32, True),
                                                       ((5, 40, 40), 512, 32, False), ((1, 3, 3), 256, 32, True)])
def test_groupnorm_silu(rows_shape, C, groups, silu):
    x = rand(*rows_shape, C, seed=5, scale=2.0) + 0.3
    g, b = rand(C, seed=6, scale=0.2) + 1, rand(C, seed=7, scale=0.2)
    xr = x.float().permute(3, 0, 1, 2).unsqueeze(0)
    ref = F.group_norm(xr, groups, g.float(), b.float(), 1e-6)
    ref = (F.silu(ref) if silu else ref)[0].permute(1, 2, 3, 0)
    got = ops.groupnorm(x.to(DEV), g.to(DEV), b.to(DEV), groups, silu=silu)
    assert rel_l2(got, ref) < 1e-3
    got2 = ops.groupnorm(x.to(DEV), g.to(DEV), b.to(DEV), groups, silu=silu)
    assert torch.equal(got, got2)                                   # two-stage statistics are deterministic


def test_gemm_f16_with_fp32_output():
    a, w = rand(300, 192, seed=40, scale=2.0), rand(328, 192, seed=41, scale=2.0)
    ref = a.double() @ w.double().T
    buf = torch.full((300, 336), 7.0, dtype=torch.float32, device=DEV)            # row-strided output, sentinel beyond N
    qkv = torch.cat([a, rand(300, 64, seed=42)], dim=1).to(DEV)                    # row-strided A
    got = ops.gemm_f16_f32out(qkv[:, :192], w.to(DEV), buf[:, :328])
    assert got.dtype == torch.float32 and rel_l2(got, ref) < 1e-6                  # fp16 products are exact in fp32
    assert (buf[:, 328:] == 7.0).all()


@pytest.mark.parametrize("T,qpf,kpf", [(3, 24, 24), (2, 10, 35), (1, 300, 300), (4, 7, 7)])
def test_softmax_fp32_logits_to_fp16_in_place(T, qpf, kpf):
    rows, sk = T * qpf, T * kpf
    cols = (sk + 63) // 64 * 64
    g = torch.Generator().manual_seed(T * 100 + qpf)
    s = torch.randn(rows, cols, generator=g) * 30.0                               # logits far beyond fp16's useful precision
    valid = (torch.arange(rows) // qpf + 1) * kpf
    mask = torch.arange(cols)[None, :] < valid[:, None]
    ref = torch.softmax((s.double() * 0.37).masked_fill(~mask, float("-inf")), dim=-1)
    row0 = 0
    p = ops.softmax_framecausal_f32_(s.to(DEV), row0, qpf, kpf, 0.37)
    assert p.dtype == torch.float16 and tuple(p.shape) == (rows, cols) and p.stride(0) == 2 * cols
    assert rel_l2(p, ref) < 1e-3
    assert (p.cpu()[~mask] == 0).all()
    # a chunk that starts inside the sequence (row0 > 0) sees the limits of its own tokens
    r0 = qpf * (T - 1) + qpf // 2
    p2 = ops.softmax_framecausal_f32_(s[r0:].contiguous().to(DEV), r0, qpf, kpf, 0.37)
    assert rel_l2(p2, ref[r0:]) < 1e-3


def test_long_rows_of_the_in_place_softmax():
    """Rows longer than one 2048-column chunk: the fp16 row overwrites fp32 logits that were read chunks earlier."""
    rows, cols = 5, 3 * 2048 + 448
    g = torch.Generator().manual_seed(3)
    s = torch.randn(rows, cols, generator=g) * 4.0
    for valid in (cols, cols - 13, 2048, 2049, 700):
        ref = torch.softmax(s[:, :valid].double(), dim=-1)
        p = ops.softmax_framecausal_f32_(s.to(DEV), 0, rows, valid, 1.0)
        assert rel_l2(p[:, :valid], ref) < 1e-3, valid
        assert p[:, valid:].abs().max().item() == 0 if valid < cols else True


@pytest.mark.parametrize("T,qpf,kpf,C", [(3, 40, 40, 128), (2, 24, 72, 64), (1, 500, 500, 512)])
def test_attention_framecausal_matches_masked_attention(T, qpf, kpf, C, monkeypatch):
    sq, sk = T * qpf, T * kpf
    sp = (sk + 63) // 64 * 64
    q, k, v = rand(sq, C, seed=50, scale=1.5), rand(sp, C, seed=51, scale=1.5), rand(sp, C, seed=52)
    k[sk:] = 0
    v[sk:] = 0
    valid = (torch.arange(sq) // qpf + 1) * kpf
    mask = torch.arange(sk)[None, :] < valid[:, None]
    s = (q.double() @ k[:sk].double().T * 0.2).masked_fill(~mask, float("-inf"))
    ref = torch.softmax(s, dim=-1) @ v[:sk].double()
    vt = v.t().contiguous().to(DEV)
    got = ops.attention_framecausal(q.to(DEV), k.to(DEV), vt, qpf, kpf, scale=0.2)
    assert rel_l2(got, ref) < 2e-3
    # the same through small chunks of query rows (a logits buffer far smaller than the problem)
    monkeypatch.setattr(ops, "VAE_ATTN_WS_BYTES", 4096)
    ops._VAE_ATTN_WS.clear()
    again = ops.attention_framecausal(q.to(DEV), k.to(DEV), vt, qpf, kpf, scale=0.2)
    ops._VAE_ATTN_WS.clear()
    assert rel_l2(again, ref) < 2e-3


def test_gemm_f16_bias_and_residual():
    a, w, b, r = rand(300, 128, seed=9), rand(72, 128, seed=10, scale=0.1), rand(72, seed=11), rand(300, 72, seed=12)
    ref = a.float() @ w.float().T + b.float()
    assert rel_l2(ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV)), ref) < 2e-3
    got = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), epilogue=ops.EPI_RESIDUAL, resid=r.to(DEV))
    assert rel_l2(got, ref + r.float()) < 2e-3


SMALL = dict(block_out_channels=(64, 64, 128, 128), norm_num_groups=16)


def build(cfg_kw, seed=0, with_encoder=False):
    cfg = VO.VAEConfig(**cfg_kw)
    P = VO.random_state_dict(cfg, seed=seed, with_encoder=with_encoder)
    P16 = {k: v.to(H16) for k, v in P.items()}
    vae = AutoencoderKLHunyuanVideo(**cfg_kw)
    vae.load_state_dict(P16, strict=True)
    return cfg, {k: v.float() for k, v in P16.items()}, vae.to(DEV, H16).eval()


@pytest.mark.parametrize("T,H,W", [(1, 8, 8), (3, 8, 6), (4, 10, 8)])
def test_decoder_tile_matches_oracle(T, H, W):
    cfg, P, vae = build(SMALL)
    z = torch.randn(1, 16, T, H, W, generator=torch.Generator().manual_seed(3))
    ref = VO.vae_decode(P, cfg, z.to(H16).float())
    got = vae_decode(z.to(DEV), vae)
    assert tuple(got.shape) == (1, 3, 4 * (T - 1) + 1, 8 * H, 8 * W) and got.dtype == H16
    assert psnr(got, ref) >= 40.0, psnr(got, ref)
    assert rel_l2(got, ref) < 1e-2


def test_framewise_decode_windows_and_crossfade_match_oracle():
    cfg, P, vae = build(SMALL, seed=1)
    z = torch.randn(1, 16, 9, 6, 6, generator=torch.Generator().manual_seed(4))     # 9 latent frames: windows at 0,3,6
    ref = VO.vae_decode(P, cfg, z.to(H16).float())
    got = vae_decode(z.to(DEV), vae)
    assert tuple(got.shape) == tuple(ref.shape) == (1, 3, 33, 48, 48)
    assert psnr(got, ref) >= 40.0, psnr(got, ref)
    vae.use_framewise_decoding = False
    whole = vae_decode(z.to(DEV), vae)
    assert psnr(whole, VO.vae_decode(P, cfg, z.to(H16).float(), framewise=False)) >= 40.0


def test_image_mode_decodes_frames_independently():
    cfg, P, vae = build(SMALL, seed=2)
    z = torch.randn(1, 16, 2, 6, 6, generator=torch.Generator().manual_seed(5))
    got = vae_decode(z.to(DEV), vae, image_mode=True)
    ref = torch.cat([VO.vae_decode(P, cfg, z[:, :, i:i + 1].to(H16).float()) for i in range(2)], dim=2)
    assert tuple(got.shape) == (1, 3, 2, 48, 48) and psnr(got, ref) >= 40.0


# ------------------------------------------------------------------ encode half (vae_encode)
@pytest.mark.parametrize("T,H,W", [(1, 64, 64), (1, 48, 80), (5, 32, 32)])
def test_encoder_moments_match_oracle(T, H, W):
    cfg, P, vae = build(SMALL, seed=3, with_encoder=True)
    x = torch.rand(1, 3, T, H, W, generator=torch.Generator().manual_seed(6)) * 2 - 1
    mean, logvar = VO.encode_moments(P, cfg, x.to(H16).float())
    dist = vae.encode(x.to(DEV, H16)).latent_dist
    To = 1 + (T - 1) // 4 if T > 1 else 1
    assert tuple(dist.mean.shape) == tuple(mean.shape) == (1, 16, mean.shape[2], H // 8, W // 8) and dist.mean.dtype == H16
    assert rel_l2(dist.mean, mean) < 1e-2 and rel_l2(dist.logvar, logvar) < 1e-2
    assert rel_l2(dist.std, torch.exp(0.5 * logvar)) < 1e-2
    assert torch.equal(dist.mode(), dist.mean)


def test_vae_encode_samples_with_the_given_noise_and_scales():
    cfg, P, vae = build(SMALL, seed=4, with_encoder=True)
    x = torch.rand(1, 3, 1, 64, 64, generator=torch.Generator().manual_seed(7)) * 2 - 1
    torch.manual_seed(11)
    got = vae_encode(x, vae)                                          # moves to vae.device / vae.dtype itself
    torch.manual_seed(11)
    noise = torch.randn(1, 16, 1, 8, 8, device=DEV, dtype=H16)       # the draw latent_dist.sample() made
    ref = VO.vae_encode(P, cfg, x.to(H16).float(), noise.float().cpu())
    assert got.dtype == H16 and tuple(got.shape) == (1, 16, 1, 8, 8)
    assert rel_l2(got, ref) < 1e-2
    # encode -> decode round trip runs end to end through both halves
    back = vae_decode(got, vae)
    assert tuple(back.shape) == (1, 3, 1, 64, 64) and torch.isfinite(back).all()


def test_encode_argument_checks_and_decoder_only_checkpoints():
    cfg, P, vae = build(SMALL, seed=5, with_encoder=True)
    with pytest.raises(RuntimeError):
        vae.encode(torch.zeros(1, 3, 1, 64, 64, dtype=H16))           # CPU tensor: no fallback
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 1, 60, 64, dtype=H16, device=DEV))
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 17, 64, 64, dtype=H16, device=DEV))
    _, _, dec_only = build(SMALL, seed=5)
    with pytest.raises(RuntimeError, match="decoder-only"):
        dec_only.encode(torch.zeros(1, 3, 1, 64, 64, dtype=H16, device=DEV))
