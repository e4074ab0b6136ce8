"""Tensor-level wrappers over the C ABI (include/framepack_hip.h).

torch is used for device memory and the current stream only: every op hands raw pointers, shapes and
strides to libframepack_hip.so.  CPU tensors are rejected: there is no fallback path.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import lib as _lib

BF16 = torch.bfloat16
F16 = torch.float16

# bench.py sets this to a list to time the dominant kernel live: every attention launch is then bracketed by
# two events recorded on the stream it is launched on (appended as (start, end, tokens, heads)).
ATTN_EVENTS: Optional[list] = None


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(*ts: Optional[torch.Tensor]) -> None:
    """Every tensor on the GPU, and on the CURRENT HIP device: a kernel launches on the current device whatever
    stream handle it is given, so a tensor of another GPU would be read through the wrong device's queue."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("framepack_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                               f"wrap the call in `with torch.cuda.device({t.device.index}):`")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor):
    """View a [.., rows, cols] tensor whose leading dims are 1 as (ptr, rows, cols, row_stride)."""
    if t.dim() > 2:
        t = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t.squeeze(0)
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"expected a [rows, cols] tensor with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    return t, t.shape[0], t.shape[1], t.stride(0)


def layernorm_modulate(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, eps: float = 1e-6,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K3: LN(x) * (1 + scale) + shift.  x [rows, cols] (row-strided ok), scale/shift [cols]."""
    _dev(x, scale, shift, out)
    x2, rows, cols, xs = _rows2d(x)
    if out is None:
        out = torch.empty((rows, cols), dtype=BF16, device=x.device)
    o2, orows, ocols, os_ = _rows2d(out)
    assert x.dtype == BF16 and out.dtype == BF16 and (orows, ocols) == (rows, cols)
    scale = scale.reshape(-1).contiguous()
    shift = shift.reshape(-1).contiguous()
    assert scale.numel() == cols and shift.numel() == cols and scale.dtype == BF16 and shift.dtype == BF16
    if rows == 0:                                   # an empty row range (a sequence-parallel rank without text rows)
        return out
    _lib.check(_lib.load().fp_layernorm_modulate_bf16(x2.data_ptr(), xs, scale.data_ptr(), shift.data_ptr(),
                                                      o2.data_ptr(), os_, rows, cols, eps, _stream(x)),
               "fp_layernorm_modulate_bf16")
    return out


def layernorm_affine(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K3, affine form: LN(x) * weight + bias (the token refiner's nn.LayerNorm).  x [rows, cols], weight/bias [cols]."""
    _dev(x, weight, bias, out)
    x2, rows, cols, xs = _rows2d(x)
    if out is None:
        out = torch.empty((rows, cols), dtype=BF16, device=x.device)
    o2, orows, ocols, os_ = _rows2d(out)
    assert x.dtype == BF16 and out.dtype == BF16 and (orows, ocols) == (rows, cols)
    weight = weight.reshape(-1).contiguous()
    bias = bias.reshape(-1).contiguous()
    assert weight.numel() == cols and bias.numel() == cols and weight.dtype == BF16 and bias.dtype == BF16
    if rows == 0:
        return out
    _lib.check(_lib.load().fp_layernorm_affine_bf16(x2.data_ptr(), xs, weight.data_ptr(), bias.data_ptr(),
                                                    o2.data_ptr(), os_, rows, cols, eps, _stream(x)),
               "fp_layernorm_affine_bf16")
    return out


def gate_residual_(x: torch.Tensor, y: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """K4: x += gate * y in place."""
    _dev(x, y, gate)
    x2, rows, cols, xs = _rows2d(x)
    y2, yr, yc, ys = _rows2d(y)
    gate = gate.reshape(-1).contiguous()
    assert (yr, yc) == (rows, cols) and gate.numel() == cols
    assert x.dtype == BF16 and y.dtype == BF16 and gate.dtype == BF16
    if rows == 0:
        return x
    _lib.check(_lib.load().fp_gate_residual_bf16(x2.data_ptr(), xs, y2.data_ptr(), ys, gate.data_ptr(), rows, cols,
                                                 _stream(x)), "fp_gate_residual_bf16")
    return x


def qk_rmsnorm_rope_(q: torch.Tensor, k: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor,
                     rope: Optional[torch.Tensor], rope_rows: int, heads: int, eps: float = 1e-6,
                     wq_txt: Optional[torch.Tensor] = None, wk_txt: Optional[torch.Tensor] = None) -> None:
    """K2: in-place RMSNorm(q), RMSNorm(k) per head (+RoPE on rows < rope_rows).
    q, k: [rows, heads*128] views sharing one row stride."""
    _dev(q, k, wq, wk, rope, wq_txt, wk_txt)
    q2, rows, cols, qs = _rows2d(q)
    k2, kr, kc, ks = _rows2d(k)
    assert (kr, kc) == (rows, cols) and qs == ks and cols == heads * 128
    assert q.dtype == BF16 and k.dtype == BF16
    for w in (wq, wk, wq_txt, wk_txt):
        assert w is None or (w.dtype == BF16 and w.numel() == 128 and w.is_contiguous())
    if rope_rows > 0:
        assert rope is not None and rope.dtype == torch.float32 and rope.is_contiguous()
        assert rope.shape[-1] == 256 and rope.numel() >= rope_rows * 256
    _lib.check(_lib.load().fp_qk_rmsnorm_rope_bf16(q2.data_ptr(), k2.data_ptr(), qs, rows, heads, 128,
                                                   wq.data_ptr(), wk.data_ptr(), _ptr(wq_txt), _ptr(wk_txt),
                                                   _ptr(rope), rope_rows, eps, _stream(q)),
               "fp_qk_rmsnorm_rope_bf16")


def qk_rmsnorm_rope_pack(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor,
                         rope: Optional[torch.Tensor], rope_rows: int, heads: int, world: int, chunks: int,
                         eps: float = 1e-6, wq_txt: Optional[torch.Tensor] = None,
                         wk_txt: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K2 for sequence parallelism: q, k, v [rows, heads*128] (one row stride) are left as they are; returns the send
    buffer of the head exchange [chunks, world, rows, 3 * hc * 128] holding RMSNorm/RoPE(q), RMSNorm/RoPE(k) and v of
    the heads each peer owns (hc = heads / world / chunks heads per chunk)."""
    _dev(q, k, v, wq, wk, rope, wq_txt, wk_txt, out)
    q2, rows, cols, qs = _rows2d(q)
    k2, kr, kc, ks = _rows2d(k)
    v2, vr, vc, vs = _rows2d(v)
    assert (kr, kc) == (rows, cols) and (vr, vc) == (rows, cols) and qs == ks == vs and cols == heads * 128
    assert q.dtype == BF16 and k.dtype == BF16 and v.dtype == BF16 and heads % (world * chunks) == 0
    for w in (wq, wk, wq_txt, wk_txt):
        assert w is None or (w.dtype == BF16 and w.numel() == 128 and w.is_contiguous())
    if rope_rows > 0:
        assert rope is not None and rope.dtype == torch.float32 and rope.is_contiguous()
        assert rope.shape[-1] == 256 and rope.numel() >= rope_rows * 256
    hwc = heads // world // chunks * 128
    if out is None:
        out = torch.empty((chunks, world, rows, 3 * hwc), dtype=BF16, device=q.device)
    assert tuple(out.shape) == (chunks, world, rows, 3 * hwc) and out.is_contiguous() and out.dtype == BF16
    _lib.check(_lib.load().fp_qk_rmsnorm_rope_pack_bf16(q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), qs, rows, heads, 128,
                                                        wq.data_ptr(), wk.data_ptr(), _ptr(wq_txt), _ptr(wk_txt),
                                                        _ptr(rope), rope_rows, eps, out.data_ptr(), world, chunks,
                                                        _stream(q)), "fp_qk_rmsnorm_rope_pack_bf16")
    return out


_ATTN_WS = {}


def _attn_workspace(device: torch.device) -> torch.Tensor:
    """Scratch for the attention kernel's split last round, one per (device, stream): calls on one stream are
    ordered, calls on different streams get different buffers (allocated once each, by the caller: the library
    itself allocates nothing)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _ATTN_WS.get(key)
    if ws is None:
        ws = torch.empty((_lib.load().fp_attn_workspace_bytes(),), dtype=torch.uint8, device=device)
        _ATTN_WS[key] = ws
    return ws


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None,
              cu_seqlens: Optional[torch.Tensor] = None, max_seqlen: Optional[int] = None,
              scale: Optional[float] = None) -> torch.Tensor:
    """K1: non-causal flash attention.  q,k,v: [tokens, heads*128] views (row-strided ok).
    cu_seqlens: int32 device [nseg+1] for varlen, None for one segment.  out may alias q."""
    _dev(q, k, v, out, cu_seqlens)
    q2, rows, cols, qs = _rows2d(q)
    k2, _, _, ks = _rows2d(k)
    v2, _, _, vs = _rows2d(v)
    assert cols == heads * 128 and k2.shape == q2.shape and v2.shape == q2.shape
    assert q.dtype == BF16 and k.dtype == BF16 and v.dtype == BF16
    if out is None:
        out = torch.empty((rows, cols), dtype=BF16, device=q.device)
    o2, orows, ocols, os_ = _rows2d(out)
    assert (orows, ocols) == (rows, cols) and out.dtype == BF16
    nseg = 1
    if cu_seqlens is not None:
        assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous()
        nseg = cu_seqlens.numel() - 1
    if max_seqlen is None:
        max_seqlen = rows
    if scale is None:
        scale = 1.0 / math.sqrt(128.0)
    ev = None
    if ATTN_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(q.device))
    ws = _attn_workspace(q.device)
    _lib.check(_lib.load().fp_attn_fwd_bf16(q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), o2.data_ptr(), qs, ks, vs, os_,
                                            _ptr(cu_seqlens), nseg, max_seqlen, rows, heads, 128, scale,
                                            ws.data_ptr(), ws.numel(), _stream(q)),
               "fp_attn_fwd_bf16")
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(q.device))
        ATTN_EVENTS.append((ev[0], ev[1], rows, heads))
    return out


_ATTN_FP8_WS = {}


def attention_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None,
                  scale: Optional[float] = None, cu_seqlens: Optional[torch.Tensor] = None,
                  max_seqlen: Optional[int] = None) -> torch.Tensor:
    """K1, fp8 variant (opt-in): Q / K / P / V in e4m3 on the fp8 MFMA; q, k, v, cu_seqlens, max_seqlen as for
    `attention` (bf16; quantised inside the call).  A few per cent away from the bf16 kernel.  out may alias q."""
    _dev(q, k, v, out, cu_seqlens)
    q2, rows, cols, qs = _rows2d(q)
    k2, _, _, ks = _rows2d(k)
    v2, _, _, vs = _rows2d(v)
    assert cols == heads * 128 and k2.shape == q2.shape and v2.shape == q2.shape
    assert q.dtype == BF16 and k.dtype == BF16 and v.dtype == BF16
    if out is None:
        out = torch.empty((rows, cols), dtype=BF16, device=q.device)
    o2, orows, ocols, os_ = _rows2d(out)
    assert (orows, ocols) == (rows, cols) and out.dtype == BF16
    nseg = 1
    if cu_seqlens is not None:
        assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous()
        nseg = cu_seqlens.numel() - 1
    if max_seqlen is None:
        max_seqlen = rows
    if scale is None:
        scale = 1.0 / math.sqrt(128.0)
    lib = _lib.load()
    need = lib.fp_attn_fp8_varlen_workspace_bytes(rows, heads, nseg, max_seqlen)
    key = (q.device, torch.cuda.current_stream(q.device).cuda_stream)
    ws = _ATTN_FP8_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _ATTN_FP8_WS[key] = torch.empty((max(need, 256),), dtype=torch.uint8, device=q.device)
    ev = None
    if ATTN_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(q.device))
    _lib.check(lib.fp_attn_fwd_fp8_varlen(q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), o2.data_ptr(), qs, ks, vs, os_,
                                          _ptr(cu_seqlens), nseg, max_seqlen, rows, heads, 128, scale, ws.data_ptr(),
                                          ws.numel(), _stream(q)), "fp_attn_fwd_fp8_varlen")
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(q.device))
        ATTN_EVENTS.append((ev[0], ev[1], rows, heads))
    return out


def silu_gemv(t: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], apply_silu: bool = True,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K7: Linear(SiLU(t)) for a single vector t [in]; weight [out, in]."""
    _dev(t, weight, bias, out)
    t = t.reshape(-1).contiguous()
    assert weight.dim() == 2 and weight.is_contiguous() and weight.shape[1] == t.numel()
    assert t.dtype == BF16 and weight.dtype == BF16 and (bias is None or bias.dtype == BF16)
    n = weight.shape[0]
    if out is None:
        out = torch.empty((n,), dtype=BF16, device=t.device)
    assert out.numel() == n and out.is_contiguous() and out.dtype == BF16
    _lib.check(_lib.load().fp_silu_gemv_bf16(t.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(),
                                             t.numel(), n, int(apply_silu), _stream(t)), "fp_silu_gemv_bf16")
    return out


def rel_l1_acc(device) -> torch.Tensor:
    """Result + scratch buffer of `rel_l1` (fp_rel_l1_acc_floats() floats)."""
    return torch.zeros((_lib.load().fp_rel_l1_acc_floats(),), dtype=torch.float32, device=device)


def rel_l1(a: torch.Tensor, b: torch.Tensor, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K10: returns a device float32 view [2] = (sum|a-b|, sum|b|); deterministic (two stages, no atomics).
    acc: a buffer from `rel_l1_acc` to reuse."""
    _dev(a, b, acc)
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() and a.dtype == BF16 and b.dtype == BF16
    if acc is None:
        acc = rel_l1_acc(a.device)
    assert acc.dtype == torch.float32 and acc.is_contiguous() and acc.numel() >= _lib.load().fp_rel_l1_acc_floats()
    _lib.check(_lib.load().fp_rel_l1_bf16(a.data_ptr(), b.data_ptr(), a.numel(), acc.data_ptr(), _stream(a)),
               "fp_rel_l1_bf16")
    return acc[:2]


EPI_BIAS, EPI_GELU, EPI_GELU_FROM, EPI_GATE_RESIDUAL, EPI_RESIDUAL = 0, 1, 2, 3, 4


def gemm(a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
         epilogue: int = EPI_BIAS, gelu_from: int = 0, resid: Optional[torch.Tensor] = None,
         gate: Optional[torch.Tensor] = None, a2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K5: out = epilogue(a @ weight.T + bias).  a [M,K] (row-strided ok), weight [N,K] (row-strided ok).
    a2: optional second K segment [M,K2] with the same row stride as a (K = K1 + K2)."""
    _dev(a, weight, bias, out, resid, gate, a2)
    a_, M, K1, lda = _rows2d(a)
    w_, N, K, ldw = _rows2d(weight)
    k1 = K1
    if a2 is not None:
        a2_, M2, K2, lda2 = _rows2d(a2)
        assert M2 == M and lda2 == lda and K1 + K2 == K
    else:
        assert K1 == K, f"A has K={K1}, weight has K={K}"
    dt = a.dtype
    assert dt in (BF16, F16) and weight.dtype == dt and (bias is None or (bias.dtype == dt and bias.numel() == N))
    if out is None:
        out = torch.empty((M, N), dtype=dt, device=a.device)
    o_, Mo, No, ldc = _rows2d(out)
    assert (Mo, No) == (M, N) and out.dtype == dt
    if M == 0:                                      # an empty row range (a sequence-parallel rank without text rows)
        return out
    ldr = 0
    rptr, gptr = None, None
    if epilogue in (EPI_GATE_RESIDUAL, EPI_RESIDUAL):
        r_, Mr, Nr, ldr = _rows2d(resid)
        assert (Mr, Nr) == (M, N) and resid.dtype == dt
        rptr = r_.data_ptr()
    if epilogue == EPI_GATE_RESIDUAL:
        gate = gate.reshape(-1).contiguous()
        assert gate.numel() == N and gate.dtype == dt
        gptr = gate.data_ptr()
    fn = _lib.load().fp_gemm_bf16 if dt == BF16 else _lib.load().fp_gemm_f16
    _lib.check(fn(a_.data_ptr(), lda, None if a2 is None else a2_.data_ptr(), k1, w_.data_ptr(), ldw,
                                        _ptr(bias), o_.data_ptr(), ldc, M, N, K, epilogue, gelu_from, rptr, ldr, gptr,
               _stream(a)), "fp_gemm_bf16" if dt == BF16 else "fp_gemm_f16")
    return out


def gemm_rows2(a: torch.Tensor, m_split: int, weight0: torch.Tensor, weight1: torch.Tensor,
               bias0: Optional[torch.Tensor], bias1: Optional[torch.Tensor], out: torch.Tensor,
               epilogue: int = EPI_BIAS, resid: Optional[torch.Tensor] = None, gate0: Optional[torch.Tensor] = None,
               gate1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K5 over two row segments in one launch: out[:m_split] = epilogue(a[:m_split] @ weight0.T + bias0) with gate0,
    out[m_split:] = epilogue(a[m_split:] @ weight1.T + bias1) with gate1 (the image and the text linear of a double-stream
    block over the rows of one buffer).  Bit-identical to the two `gemm` calls.  bf16; epilogues BIAS, GELU, GATE_RESIDUAL."""
    _dev(a, weight0, weight1, bias0, bias1, out, resid, gate0, gate1)
    a_, M, K, lda = _rows2d(a)
    w0_, N, K0, ldw = _rows2d(weight0)
    w1_, N1, K1, ldw1 = _rows2d(weight1)
    assert (N1, K1, ldw1) == (N, K0, ldw) and K0 == K and 0 <= m_split <= M
    assert a.dtype == BF16 and weight0.dtype == BF16 and weight1.dtype == BF16 and out.dtype == BF16
    assert (bias0 is None) == (bias1 is None) and (bias0 is None or (bias0.numel() == N and bias1.numel() == N))
    o_, Mo, No, ldc = _rows2d(out)
    assert (Mo, No) == (M, N)
    if M == 0:
        return out
    ldr, rptr, g0, g1 = 0, None, None, None
    if epilogue == EPI_GATE_RESIDUAL:
        r_, Mr, Nr, ldr = _rows2d(resid)
        assert (Mr, Nr) == (M, N) and resid.dtype == BF16
        rptr = r_.data_ptr()
        gate0, gate1 = gate0.reshape(-1).contiguous(), gate1.reshape(-1).contiguous()
        assert gate0.numel() == N and gate1.numel() == N and gate0.dtype == BF16 and gate1.dtype == BF16
        g0, g1 = gate0.data_ptr(), gate1.data_ptr()
    else:
        assert epilogue in (EPI_BIAS, EPI_GELU)
    _lib.check(_lib.load().fp_gemm_bf16_rows2(a_.data_ptr(), lda, w0_.data_ptr(), w1_.data_ptr(), ldw, _ptr(bias0),
                                              _ptr(bias1), o_.data_ptr(), ldc, M, m_split, N, K, epilogue, 0, rptr, ldr,
                                              g0, g1, _stream(a)), "fp_gemm_bf16_rows2")
    return out


def gemm_f32out(a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K5 with fp32 output: out [M,N] fp32 = a [M,K] @ weight [N,K].T + bias, a / weight bf16, bias fp32 [N] or None
    (the fp32 head of `high_quality_fp32_output_for_inference`)."""
    _dev(a, weight, bias, out)
    a_, M, K, lda = _rows2d(a)
    w_, N, Kw, ldw = _rows2d(weight)
    assert K == Kw and a.dtype == BF16 and weight.dtype == BF16
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous())
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    o_, Mo, No, ldc = _rows2d(out)
    assert (Mo, No) == (M, N) and out.dtype == torch.float32
    if M == 0:
        return out
    _lib.check(_lib.load().fp_gemm_bf16_f32out(a_.data_ptr(), lda, w_.data_ptr(), ldw, _ptr(bias), o_.data_ptr(), ldc,
                                               M, N, K, _stream(a)), "fp_gemm_bf16_f32out")
    return out


def gemm_qknorm(a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, wq: torch.Tensor,
                wk: torch.Tensor, rope: Optional[torch.Tensor], rope_rows: int, qcol0: int, kcol0: int, heads: int,
                epilogue: int = EPI_BIAS, gelu_from: int = 0, eps: float = 1e-6, wq_txt: Optional[torch.Tensor] = None,
                wk_txt: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K5 + K2 in one launch: out = epilogue(a @ weight.T + bias) with the per-head RMSNorm of the q and k columns
    ([qcol0, +heads*128), [kcol0, +heads*128)) and RoPE on rows < rope_rows applied in the GEMM's epilogue - the same
    bits as gemm(...) followed by qk_rmsnorm_rope_(...) on those column views.  bf16; out 16-byte aligned, row stride
    % 8 == 0 (`gemm_qknorm_ok`)."""
    _dev(a, weight, bias, out, wq, wk, rope, wq_txt, wk_txt)
    a_, M, K1, lda = _rows2d(a)
    w_, N, K, ldw = _rows2d(weight)
    o_, Mo, No, ldc = _rows2d(out)
    assert K1 == K and (Mo, No) == (M, N) and a.dtype == BF16 and weight.dtype == BF16 and out.dtype == BF16
    assert bias is None or (bias.dtype == BF16 and bias.numel() == N)
    assert epilogue in (EPI_BIAS, EPI_GELU_FROM) and gemm_qknorm_ok(out)
    for w in (wq, wk, wq_txt, wk_txt):
        assert w is None or (w.dtype == BF16 and w.numel() == 128 and w.is_contiguous())
    if rope_rows > 0:
        assert rope is not None and rope.dtype == torch.float32 and rope.is_contiguous()
        assert rope.shape[-1] == 256 and rope.numel() >= rope_rows * 256
    if M == 0:
        return out
    _lib.check(_lib.load().fp_gemm_bf16_qknorm(a_.data_ptr(), lda, w_.data_ptr(), ldw, _ptr(bias), o_.data_ptr(), ldc, M, N,
                                               K, epilogue, gelu_from, wq.data_ptr(), wk.data_ptr(), _ptr(wq_txt),
                                               _ptr(wk_txt), _ptr(rope), rope_rows, qcol0, kcol0, heads, 128, eps,
                                               _stream(a)), "fp_gemm_bf16_qknorm")
    return out


def gemm_qknorm_ok(out: torch.Tensor) -> bool:
    """The fused form runs on the 4-wave kernel only: output rows stored 16 bytes per lane."""
    return out.dtype == BF16 and out.data_ptr() % 16 == 0 and out.stride(-2) % 8 == 0 and out.stride(-1) == 1


# ---- K6: fp8 linears on the fp8 MFMA (opt-in: e4m3 activations are outside the 1e-2 parity bar) -------------

FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def quantize_weight_fp8(weight: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """Host-side weight preparation: [N,K] -> (e4m3 [N,K], fp32 scale [N]).  With `scale` (the per-tensor
    `scale_weight` of a reference fp8 checkpoint, ref utils/fp8_optimization_utils.py) the weight is already
    e4m3 and the scale is only broadcast; otherwise one scale per output channel is derived (amax / 448)."""
    N = weight.shape[0]
    if weight.dtype == FP8:
        assert scale is not None
        return weight.contiguous(), scale.float().reshape(-1).expand(N).contiguous()
    w = weight.float()
    sc = (w.abs().amax(dim=1) / FP8_MAX).clamp_min(1e-12)
    return (w / sc[:, None]).to(FP8).contiguous(), sc.contiguous()


def quantize_rows_fp8(x: torch.Tensor, x2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                      scale: Optional[torch.Tensor] = None):
    """K6a: bf16 [M,K1] (+ [M,K2] side by side in K) -> (e4m3 [M,K], fp32 scale [M]); one scale per token row."""
    _dev(x, x2, out, scale)
    x_, M, K1, ld1 = _rows2d(x)
    assert x.dtype == BF16
    if M == 0:
        K2 = 0 if x2 is None else x2.shape[-1]
        return (torch.empty((0, K1 + K2), dtype=FP8, device=x.device) if out is None else out,
                torch.empty((0,), dtype=torch.float32, device=x.device) if scale is None else scale)
    K2, ld2, x2p = 0, 0, None
    if x2 is not None:
        x2_, M2, K2, ld2 = _rows2d(x2)
        assert M2 == M and x2.dtype == BF16
        x2p = x2_.data_ptr()
    if out is None:
        out = torch.empty((M, K1 + K2), dtype=FP8, device=x.device)
    if scale is None:
        scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    assert out.dtype == FP8 and out.shape == (M, K1 + K2) and out.stride(1) == 1
    assert scale.dtype == torch.float32 and scale.numel() == M and scale.is_contiguous()
    _lib.check(_lib.load().fp_quantize_rows_fp8(x_.data_ptr(), ld1, K1, x2p, ld2, K2, out.data_ptr(), out.stride(0),
                                                scale.data_ptr(), M, _stream(x)), "fp_quantize_rows_fp8")
    return out, scale


def layernorm_modulate_fp8(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, eps: float = 1e-6,
                           out: Optional[torch.Tensor] = None, out_scale: Optional[torch.Tensor] = None):
    """K3 + K6a in one pass: (e4m3 [rows, cols], fp32 scale [rows]) of LN(x) * (1 + scale) + shift - what
    `quantize_rows_fp8(layernorm_modulate(x, scale, shift))` returns, bit for bit."""
    _dev(x, scale, shift, out, out_scale)
    x2, rows, cols, xs = _rows2d(x)
    assert x.dtype == BF16
    if out is None:
        out = torch.empty((rows, cols), dtype=FP8, device=x.device)
    if out_scale is None:
        out_scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    assert out.dtype == FP8 and tuple(out.shape) == (rows, cols) and (rows == 0 or out.stride(1) == 1)
    assert out_scale.dtype == torch.float32 and out_scale.numel() == rows and out_scale.is_contiguous()
    scale = scale.reshape(-1).contiguous()
    shift = shift.reshape(-1).contiguous()
    assert scale.numel() == cols and shift.numel() == cols and scale.dtype == BF16 and shift.dtype == BF16
    if rows == 0:
        return out, out_scale
    _lib.check(_lib.load().fp_layernorm_modulate_fp8(x2.data_ptr(), xs, scale.data_ptr(), shift.data_ptr(),
                                                     out.data_ptr(), out.stride(0), out_scale.data_ptr(), rows, cols,
                                                     eps, _stream(x)), "fp_layernorm_modulate_fp8")
    return out, out_scale


def gemm_fp8(a_q: torch.Tensor, a_scale: torch.Tensor, w_q: torch.Tensor, w_scale: torch.Tensor,
             bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, epilogue: int = EPI_BIAS,
             gelu_from: int = 0, resid: Optional[torch.Tensor] = None,
             gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K6b: out bf16 = epilogue(a_scale[m] * w_scale[n] * (a_q @ w_q.T) + bias); a_q [M,K], w_q [N,K] e4m3."""
    _dev(a_q, a_scale, w_q, w_scale, bias, out, resid, gate)
    M, K = a_q.shape
    N, Kw = w_q.shape
    assert a_q.dtype == FP8 and w_q.dtype == FP8 and K == Kw and a_q.stride(1) == 1 and w_q.stride(1) == 1
    assert a_scale.dtype == torch.float32 and a_scale.numel() == M and a_scale.is_contiguous()
    assert w_scale.dtype == torch.float32 and w_scale.numel() == N and w_scale.is_contiguous()
    assert bias is None or (bias.dtype == BF16 and bias.numel() == N)
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a_q.device)
    o_, Mo, No, ldc = _rows2d(out)
    assert (Mo, No) == (M, N) and out.dtype == BF16
    if M == 0:
        return out
    ldr, rptr, gptr = 0, None, None
    if epilogue in (EPI_GATE_RESIDUAL, EPI_RESIDUAL):
        r_, Mr, Nr, ldr = _rows2d(resid)
        assert (Mr, Nr) == (M, N) and resid.dtype == BF16
        rptr = r_.data_ptr()
    if epilogue == EPI_GATE_RESIDUAL:
        gate = gate.reshape(-1).contiguous()
        assert gate.numel() == N and gate.dtype == BF16
        gptr = gate.data_ptr()
    _lib.check(_lib.load().fp_gemm_fp8_bf16(a_q.data_ptr(), a_q.stride(0), a_scale.data_ptr(), w_q.data_ptr(),
                                            w_q.stride(0), w_scale.data_ptr(), _ptr(bias), o_.data_ptr(), ldc, M, N, K,
                                            epilogue, gelu_from, rptr, ldr, gptr, _stream(a_q)), "fp_gemm_fp8_bf16")
    return out


# ---- K11: causal-3D VAE decode (fp16, channels-last volumes) ------------------------------------------------

def conv3d_causal(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                  up: Tuple[int, int, int] = (1, 1, 1), resid: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal 3x3x3 convolution with replicate padding.  x [T,H,W,Cin] fp16 contiguous, weight [Cout, 27*Cin]
    (tap-major), `up` = nearest-upsampling factors applied on the fly to x (first frame not doubled in time)."""
    _dev(x, weight, bias, resid, out)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == F16 and weight.dtype == F16 and weight.is_contiguous()
    Ti, Hi, Wi, Cin = x.shape
    Cout = weight.shape[0]
    assert weight.shape[1] == 27 * Cin
    To = 2 * Ti - 1 if up[0] == 2 else Ti
    shape = (To, Hi * up[1], Wi * up[2], Cout)
    if out is None:
        out = torch.empty(shape, dtype=F16, device=x.device)
    assert tuple(out.shape) == shape and out.is_contiguous() and out.dtype == F16
    if resid is not None:
        assert tuple(resid.shape) == shape and resid.is_contiguous() and resid.dtype == F16
    if bias is not None:
        assert bias.dtype == F16 and bias.numel() == Cout and bias.is_contiguous()
    _lib.check(_lib.load().fp_conv3d_causal_f16(x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), Ti, Hi, Wi,
                                                up[0], up[1], up[2], Cin, Cout, _ptr(resid), _stream(x)),
               "fp_conv3d_causal_f16")
    return out


def conv3d_causal_strided(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                          stride: Tuple[int, int, int]) -> torch.Tensor:
    """The same convolution with stride 1 or 2 per axis (the encoder's downsamplers): output voxel o reads input
    voxels s*o-2..s*o in time and s*o-1..s*o+1 in space, clamped into the volume."""
    _dev(x, weight, bias)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == F16 and weight.dtype == F16 and weight.is_contiguous()
    Ti, Hi, Wi, Cin = x.shape
    Cout = weight.shape[0]
    assert weight.shape[1] == 27 * Cin
    out = torch.empty(((Ti - 1) // stride[0] + 1, (Hi - 1) // stride[1] + 1, (Wi - 1) // stride[2] + 1, Cout),
                      dtype=F16, device=x.device)
    if bias is not None:
        assert bias.dtype == F16 and bias.numel() == Cout and bias.is_contiguous()
    _lib.check(_lib.load().fp_conv3d_causal_strided_f16(x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), Ti, Hi,
                                                        Wi, stride[0], stride[1], stride[2], Cin, Cout, _stream(x)),
               "fp_conv3d_causal_strided_f16")
    return out


_GN_WS = {}


def groupnorm_stats(x: torch.Tensor, groups: int, eps: float = 1e-6, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm statistics of a channels-last volume x [..., C] fp16: fp32 [2 * groups] = (mean, rstd) per group,
    two deterministic stages.  The partials' workspace is kept per (device, stream, C): no allocation per call."""
    _dev(x, stats)
    assert x.is_contiguous() and x.dtype == F16
    C = x.shape[-1]
    rows = x.numel() // C
    lib = _lib.load()
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream, C)
    ws = _GN_WS.get(key)
    if ws is None:
        ws = _GN_WS[key] = torch.empty((lib.fp_groupnorm_workspace_floats(C),), dtype=torch.float32, device=x.device)
    if stats is None:
        stats = torch.empty((2 * groups,), dtype=torch.float32, device=x.device)
    assert stats.dtype == torch.float32 and stats.numel() == 2 * groups and stats.is_contiguous()
    _lib.check(lib.fp_groupnorm_stats_f16(x.data_ptr(), rows, C, groups, eps, ws.data_ptr(), stats.data_ptr(),
                                          _stream(x)), "fp_groupnorm_stats_f16")
    return stats


def groupnorm_apply(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int,
                    silu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = f(GroupNorm(x; stats) * gamma + beta), f = SiLU or identity; out may be x."""
    _dev(x, stats, gamma, beta, out)
    assert x.is_contiguous() and x.dtype == F16 and gamma.dtype == F16 and beta.dtype == F16
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.is_contiguous() and out.dtype == F16
    _lib.check(_lib.load().fp_groupnorm_apply_f16(x.data_ptr(), out.data_ptr(), rows, C, groups, stats.data_ptr(),
                                                  gamma.data_ptr(), beta.data_ptr(), int(silu), _stream(x)),
               "fp_groupnorm_apply_f16")
    return out


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float = 1e-6,
              silu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm (+SiLU) over a channels-last volume x [..., C] fp16 (all leading dims are the group's extent)."""
    return groupnorm_apply(x, groupnorm_stats(x, groups, eps), gamma, beta, groups, silu, out)


_VAE_ATTN_WS = {}
VAE_ATTN_WS_BYTES = 512 << 20          # fp32 logits of one query chunk (the whole S x S matrix is never materialised)


def gemm_f16_f32out(a: torch.Tensor, weight: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out [M,N] fp32 = a [M,K] @ weight [N,K].T, fp16 operands (row-strided ok), no bias."""
    _dev(a, weight, out)
    a_, M, K, lda = _rows2d(a)
    w_, N, Kw, ldw = _rows2d(weight)
    o_, Mo, No, ldc = _rows2d(out)
    assert K == Kw and (Mo, No) == (M, N) and a.dtype == F16 and weight.dtype == F16 and out.dtype == torch.float32
    _lib.check(_lib.load().fp_gemm_f16_f32out(a_.data_ptr(), lda, w_.data_ptr(), ldw, o_.data_ptr(), ldc, M, N, K,
                                              _stream(a)), "fp_gemm_f16_f32out")
    return out


def softmax_framecausal_f32_(scores: torch.Tensor, row0: int, q_per_frame: int, k_per_frame: int,
                             scale: float) -> torch.Tensor:
    """Frame-causal softmax of fp32 logits [rows, cols] (contiguous); returns the fp16 probabilities, written over the
    front of each row: a view [rows, cols] with row stride 2 * cols.  Query i is token row0 + i and keeps the keys
    below ((row0 + i) // q_per_frame + 1) * k_per_frame; the other columns become 0."""
    _dev(scores)
    assert scores.dim() == 2 and scores.is_contiguous() and scores.dtype == torch.float32
    rows, cols = scores.shape
    _lib.check(_lib.load().fp_softmax_framecausal_f32_f16(scores.data_ptr(), rows, cols, cols, row0, q_per_frame,
                                                          k_per_frame, scale, _stream(scores)),
               "fp_softmax_framecausal_f32_f16")
    return scores.view(F16)[:, :cols]


def attention_framecausal(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, q_per_frame: int, k_per_frame: int,
                          out: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
    """The VAE mid-block's attention (one head of C channels) under the frame-causal mask: query token i (frame
    i // q_per_frame) attends to the keys below (frame + 1) * k_per_frame.
      q [Sq, C], k [>= Sk rounded up to 64 rows, C] (row-strided ok), vt [C, >= Sk rounded up to 64] = V transposed with
      FINITE padding columns, all fp16; Sq = T * q_per_frame, Sk = T * k_per_frame.
    Per frame and chunk of query rows: fp32 logits (fp_gemm_f16_f32out) against the keys the frame may see only, softmax
    into fp16 in place, P.V - the logits buffer is VAE_ATTN_WS_BYTES at most, cached per (device, stream)."""
    _dev(q, k, vt, out)
    Sq, C = q.shape
    T = Sq // q_per_frame
    assert Sq == T * q_per_frame and q.dtype == F16 and k.dtype == F16 and vt.dtype == F16
    n_max = (T * k_per_frame + 63) // 64 * 64
    assert k.shape[0] >= n_max and k.shape[1] == C and vt.shape[0] == C and vt.shape[1] >= n_max
    if out is None:
        out = torch.empty((Sq, C), dtype=F16, device=q.device)
    key = (q.device, torch.cuda.current_stream(q.device).cuda_stream)
    ws = _VAE_ATTN_WS.get(key)
    need = min(Sq * n_max, max(VAE_ATTN_WS_BYTES // 4, 256 * n_max))
    if ws is None or ws.numel() < need:
        ws = _VAE_ATTN_WS[key] = torch.empty((need,), dtype=torch.float32, device=q.device)
    for f in range(T):
        n = ((f + 1) * k_per_frame + 63) // 64 * 64
        rows_max = max(256, ws.numel() // n // 256 * 256)
        for r0 in range(f * q_per_frame, (f + 1) * q_per_frame, rows_max):
            r1 = min(r0 + rows_max, (f + 1) * q_per_frame)
            logits = ws[:(r1 - r0) * n].view(r1 - r0, n)
            gemm_f16_f32out(q[r0:r1], k[:n], logits)
            p = softmax_framecausal_f32_(logits, r0, q_per_frame, k_per_frame, scale)
            gemm(p, vt[:, :n], None, out=out[r0:r1])
    return out
