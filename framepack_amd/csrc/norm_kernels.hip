// HBM-bound fused kernels of the packed DiT (K2, K3, K4, K7, K10), gfx950 / wave64.
// All of them move 16 B per lane per access and reduce with wave shuffles; none stages through LDS
// (there is no reuse to stage).  C-ABI entry points at the bottom; declarations in
// include/framepack_hip.h.
#include "fp_common.h"

// ---------------------------------------------------------------------------------------------
// K3  y = LayerNorm(x) * (1 + scale) + shift      (no affine, fp32 statistics, one rounding)
//     replaces ref dit_common.py:9-10 + hunyuan_video_packed.py:453,476,502,642-643
//     One wave per row; the row lives in registers between the statistics and the write, so the
//     algorithmic traffic is one read + one write of the row.
//     AFFINE: y = LayerNorm(x) * scale + shift - the token refiner's nn.LayerNorm(elementwise_affine=True) with
//     scale = weight, shift = bias (ref hunyuan_video_packed.py:264, 273 through dit_common.py:9-10: F.layer_norm in the
//     input dtype = fp32 arithmetic, one rounding).
// ---------------------------------------------------------------------------------------------
template <int NCH, bool AFFINE = false>   // NCH chunks of 512 elements per row (cols <= NCH*512)
__global__ __launch_bounds__(256) void ln_modulate_kernel(const bf16* __restrict__ x, int64_t x_rs,
                                                          const bf16* __restrict__ scale,
                                                          const bf16* __restrict__ shift,
                                                          bf16* __restrict__ y, int64_t y_rs, int rows, int cols,
                                                          float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + (int64_t)row * x_rs;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < cols) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = (float)t[j];
                s += v[c][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)cols;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < cols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[c][j] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)cols + eps);
    bf16* yr = y + (int64_t)row * y_rs;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < cols) {
            const bf16x8 sc = *reinterpret_cast<const bf16x8*>(scale + col);
            const bf16x8 sh = *reinterpret_cast<const bf16x8*>(shift + col);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = (bf16)((v[c][j] - mean) * rstd * ((AFFINE ? 0.f : 1.f) + (float)sc[j]) + (float)sh[j]);
            *reinterpret_cast<bf16x8*>(yr + col) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4  x += gate * y     (ref hunyuan_video_packed.py:636-637, 649-650, 570-571)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_residual_kernel(bf16* __restrict__ x, int64_t x_rs,
                                                            const bf16* __restrict__ y, int64_t y_rs,
                                                            const bf16* __restrict__ gate, int rows, int cols) {
    const int cpr = cols >> 3;                       // 16-B chunks per row
    const int64_t total = (int64_t)rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / cpr), col = (int)(i % cpr) * 8;
        bf16x8* xp = reinterpret_cast<bf16x8*>(x + (int64_t)row * x_rs + col);
        const bf16x8 xv = *xp;
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + (int64_t)row * y_rs + col);
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(gate + col);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)xv[j] + (float)gv[j] * (float)yv[j]);
        *xp = o;
    }
}

// ---------------------------------------------------------------------------------------------
// K2  in-place RMSNorm(q), RMSNorm(k) over head_dim = 128 (+ RoPE on the first rope_rows tokens)
//     replaces ref dit_common.py:31-39 + hunyuan_video_packed.py:80-86 (and the cats of :180-181)
//     16 lanes per head-row (8 elements = 4 rotary pairs per lane).  Rounding points follow the
//     reference: bf16 after the normalisation, bf16 after the weight, bf16 after the rotation.
//     Rows >= rope_rows (the text tokens) use the second weight pair when one is given
//     (norm_added_q / norm_added_k of the double-stream blocks).
//     PACK (Ulysses sequence parallelism): q, k and v are read-only and the results - v copied as it is - go to
//     `pack`, the send buffer of the head exchange, laid out [chunk][peer][row][q|k|v][heads per chunk x 128]: head h
//     belongs to peer h / hl, chunk (h % hl) / hc.  The exchange then needs no packing pass of its own.
// ---------------------------------------------------------------------------------------------
template <bool PACK>
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_kernel(bf16* __restrict__ q, bf16* __restrict__ k,
                                                              int64_t row_stride, int rows, int heads,
                                                              const bf16* __restrict__ wq,
                                                              const bf16* __restrict__ wk,
                                                              const bf16* __restrict__ wq_txt,
                                                              const bf16* __restrict__ wk_txt,
                                                              const float* __restrict__ rope, int rope_rows,
                                                              float eps, const bf16* __restrict__ v,
                                                              bf16* __restrict__ pack, int hl, int hc) {
    const int sub = threadIdx.x & 15;                        // 16-B chunk of the 128-wide head row
    const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // (row, which, head)
    const int per_row = (PACK ? 3 : 2) * heads;
    const int64_t nitems = (int64_t)rows * per_row;
    if (item >= nitems) return;                              // whole 16-lane groups leave together
    const int row = (int)(item / per_row);
    const int rem = (int)(item % per_row);
    const int which = rem / heads, head = rem % heads;
    bf16* base = (which == 0 ? q : (which == 1 ? k : const_cast<bf16*>(v))) + (int64_t)row * row_stride + head * 128 + sub * 8;
    bf16* dst = base;
    if (PACK) {
        const int peer = head / hl, chunk = (head % hl) / hc, hi = head % hc;
        const int world = heads / hl;
        dst = pack + ((((int64_t)chunk * world + peer) * rows + row) * 3 + which) * (hc * 128) + hi * 128 + sub * 8;
        if (which == 2) {
            *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(base);
            return;
        }
    }
    const bool is_txt = row >= rope_rows;
    const bf16* w = which ? ((is_txt && wk_txt) ? wk_txt : wk) : ((is_txt && wq_txt) ? wq_txt : wq);

    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(base);
    float f[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f[j] = (float)xv[j];
        ss += f[j] * f[j];
    }
    const float r = rsqrtf(group16_sum(ss) * (1.f / 128.f) + eps);
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + sub * 8);
    float n[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bf16 a = (bf16)(f[j] * r);                     // .to(input_dtype)
        n[j] = (float)(bf16)((float)a * (float)wv[j]);       // * weight.to(input_dtype)
    }
    bf16x8 o;
    if (!is_txt) {
        const float* cs = rope + (int64_t)row * 256 + sub * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), c1 = *reinterpret_cast<const f32x4*>(cs + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + 128), s1 = *reinterpret_cast<const f32x4*>(cs + 132);
        const float c[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
        for (int p = 0; p < 4; ++p) {                        // pair (2p, 2p+1): rot = (-imag, real)
            o[2 * p] = (bf16)(n[2 * p] * c[2 * p] + (-n[2 * p + 1]) * s[2 * p]);
            o[2 * p + 1] = (bf16)(n[2 * p + 1] * c[2 * p + 1] + n[2 * p] * s[2 * p + 1]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)n[j];
    }
    *reinterpret_cast<bf16x8*>(dst) = o;
}

// ---------------------------------------------------------------------------------------------
// K7  out[n] = sum_k W[n,k] * silu(t[k]) + b[n]     (batch-1 modulation GEMV: pure weight stream)
//     replaces ref hunyuan_video_packed.py:451, 474, 500, 245.  Each wave keeps silu(t) in
//     registers and streams ROWS_PER_WAVE weight rows straight to VGPRs (no LDS: nothing is shared).
// ---------------------------------------------------------------------------------------------
template <int NCH, int RPW>
__global__ __launch_bounds__(256) void silu_gemv_kernel(const bf16* __restrict__ t, const bf16* __restrict__ W,
                                                        const bf16* __restrict__ b, bf16* __restrict__ out,
                                                        int in_dim, int out_dim, int apply_silu) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= out_dim) return;
    float a[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < in_dim) {
            const bf16x8 tv = *reinterpret_cast<const bf16x8*>(t + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (float)tv[j];
                // SiLU result is rounded to bf16 as the reference's nn.SiLU on a bf16 tensor does
                a[c][j] = apply_silu ? (float)(bf16)(v / (1.f + __expf(-v))) : v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[c][j] = 0.f;
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = row0 + rr;
        if (row >= out_dim) break;
        const bf16* wr = W + (int64_t)row * in_dim;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < in_dim) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wr + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += a[c][j] * (float)wv[j];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) out[row] = (bf16)(acc + (b ? (float)b[row] : 0.f));
    }
}

// ---------------------------------------------------------------------------------------------
// K10  TeaCache relative-L1 partial sums: acc[0] += sum|a-b|, acc[1] += sum|b|   (fp32 atomics)
//      replaces ref hunyuan_video_packed.py:957-958 (two mean() passes + abs/sub temporaries)
// ---------------------------------------------------------------------------------------------
// Two deterministic stages (no atomics: the TeaCache decision must not depend on the order in which blocks retire):
// every block writes its partial (sum |a - b|, sum |b|) to part[2 * block ..], then one wave adds the partials in
// a fixed order.
constexpr int REL_L1_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(256) void rel_l1_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                     int64_t n8, float* __restrict__ part) {
    float d = 0.f, m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 av = reinterpret_cast<const bf16x8*>(a)[i];
        const bf16x8 bv = reinterpret_cast<const bf16x8*>(b)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // the reference subtracts in bf16 (rounding the difference) before abs().mean()
            d += fabsf((float)(bf16)((float)av[j] - (float)bv[j]));
            m += fabsf((float)bv[j]);
        }
    }
    d = wave_sum(d);
    m = wave_sum(m);
    __shared__ float sh[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = d;
        sh[1][wv] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
__global__ __launch_bounds__(64) void rel_l1_final_kernel(const float* __restrict__ part, int nblocks,
                                                          float* __restrict__ acc) {
    float d = 0.f, m = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        d += part[2 * i];
        m += part[2 * i + 1];
    }
    d = wave_sum(d);
    m = wave_sum(m);
    if (threadIdx.x == 0) {
        acc[0] = d;
        acc[1] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool AFFINE>
static int ln_launch(const void* x, int64_t x_row_stride, const void* scale, const void* shift, void* y,
                     int64_t y_row_stride, int rows, int cols, float eps, void* stream) {
    FP_REQUIRE(x && scale && shift && y && rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096);
    FP_REQUIRE(x_row_stride % 8 == 0 && y_row_stride % 8 == 0 && x_row_stride >= cols && y_row_stride >= cols);
    FP_REQUIRE(aligned16(x) && aligned16(y) && aligned16(scale) && aligned16(shift));
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const bf16 *xp = (const bf16*)x, *sc = (const bf16*)scale, *sh = (const bf16*)shift;
    bf16* yp = (bf16*)y;
    if (cols <= 512)
        ln_modulate_kernel<1, AFFINE><<<grid, block, 0, st>>>(xp, x_row_stride, sc, sh, yp, y_row_stride, rows, cols, eps);
    else if (cols <= 3072)
        ln_modulate_kernel<6, AFFINE><<<grid, block, 0, st>>>(xp, x_row_stride, sc, sh, yp, y_row_stride, rows, cols, eps);
    else
        ln_modulate_kernel<8, AFFINE><<<grid, block, 0, st>>>(xp, x_row_stride, sc, sh, yp, y_row_stride, rows, cols, eps);
    FP_CHECK_LAUNCH();
    return 0;
}

extern "C" int fp_layernorm_modulate_bf16(const void* x, int64_t x_row_stride, const void* scale, const void* shift,
                                          void* y, int64_t y_row_stride, int rows, int cols, float eps,
                                          void* stream) {
    return ln_launch<false>(x, x_row_stride, scale, shift, y, y_row_stride, rows, cols, eps, stream);
}

extern "C" int fp_layernorm_affine_bf16(const void* x, int64_t x_row_stride, const void* weight, const void* bias,
                                        void* y, int64_t y_row_stride, int rows, int cols, float eps, void* stream) {
    return ln_launch<true>(x, x_row_stride, weight, bias, y, y_row_stride, rows, cols, eps, stream);
}

extern "C" int fp_gate_residual_bf16(void* x, int64_t x_row_stride, const void* y, int64_t y_row_stride,
                                     const void* gate, int rows, int cols, void* stream) {
    FP_REQUIRE(x && y && gate && rows >= 0 && cols > 0 && cols % 8 == 0);
    FP_REQUIRE(x_row_stride % 8 == 0 && y_row_stride % 8 == 0 && x_row_stride >= cols && y_row_stride >= cols);
    FP_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gate));
    if (rows == 0) return 0;
    const int64_t total = (int64_t)rows * (cols / 8);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    gate_residual_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((bf16*)x, x_row_stride, (const bf16*)y, y_row_stride,
                                                                 (const bf16*)gate, rows, cols);
    FP_CHECK_LAUNCH();
    return 0;
}

extern "C" int fp_qk_rmsnorm_rope_bf16(void* q, void* k, int64_t row_stride, int rows, int heads, int head_dim,
                                       const void* wq, const void* wk, const void* wq_txt, const void* wk_txt,
                                       const float* rope, int rope_rows, float eps, void* stream) {
    FP_REQUIRE(q && k && wq && wk && rows >= 0 && heads > 0 && head_dim == 128);
    FP_REQUIRE(row_stride % 8 == 0 && row_stride >= (int64_t)heads * 128);
    FP_REQUIRE(rope_rows >= 0 && rope_rows <= rows && (rope_rows == 0 || rope));
    FP_REQUIRE(aligned16(q) && aligned16(k) && aligned16(wq) && aligned16(wk) && aligned16(rope));
    FP_REQUIRE((!wq_txt || aligned16(wq_txt)) && (!wk_txt || aligned16(wk_txt)));
    if (rows == 0) return 0;
    const int64_t items = (int64_t)rows * 2 * heads;
    qk_rmsnorm_rope_kernel<false><<<(unsigned)((items + 15) / 16), 256, 0, (hipStream_t)stream>>>(
        (bf16*)q, (bf16*)k, row_stride, rows, heads, (const bf16*)wq, (const bf16*)wk, (const bf16*)wq_txt,
        (const bf16*)wk_txt, rope, rope_rows, eps, nullptr, nullptr, heads, heads);
    FP_CHECK_LAUNCH();
    return 0;
}

extern "C" int fp_qk_rmsnorm_rope_pack_bf16(const void* q, const void* k, const void* v, int64_t row_stride, int rows,
                                            int heads, int head_dim, const void* wq, const void* wk,
                                            const void* wq_txt, const void* wk_txt, const float* rope, int rope_rows,
                                            float eps, void* pack, int world, int chunks, void* stream) {
    FP_REQUIRE(q && k && v && pack && wq && wk && rows >= 0 && heads > 0 && head_dim == 128);
    FP_REQUIRE(world >= 1 && chunks >= 1 && heads % (world * chunks) == 0);
    FP_REQUIRE(row_stride % 8 == 0 && row_stride >= (int64_t)heads * 128);
    FP_REQUIRE(rope_rows >= 0 && rope_rows <= rows && (rope_rows == 0 || rope));
    FP_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(pack) && aligned16(wq) && aligned16(wk) &&
               aligned16(rope));
    FP_REQUIRE((!wq_txt || aligned16(wq_txt)) && (!wk_txt || aligned16(wk_txt)));
    if (rows == 0) return 0;
    const int64_t items = (int64_t)rows * 3 * heads;
    const int hl = heads / world, hc = hl / chunks;
    qk_rmsnorm_rope_kernel<true><<<(unsigned)((items + 15) / 16), 256, 0, (hipStream_t)stream>>>(
        const_cast<bf16*>((const bf16*)q), const_cast<bf16*>((const bf16*)k), row_stride, rows, heads, (const bf16*)wq,
        (const bf16*)wk, (const bf16*)wq_txt, (const bf16*)wk_txt, rope, rope_rows, eps, (const bf16*)v, (bf16*)pack,
        hl, hc);
    FP_CHECK_LAUNCH();
    return 0;
}

extern "C" int fp_silu_gemv_bf16(const void* t, const void* W, const void* bias, void* out, int in_dim, int out_dim,
                                 int apply_silu, void* stream) {
    FP_REQUIRE(t && W && out && in_dim > 0 && in_dim % 8 == 0 && in_dim <= 4096 && out_dim > 0);
    FP_REQUIRE(aligned16(t) && aligned16(W));
    hipStream_t st = (hipStream_t)stream;
    constexpr int RPW = 4;
    dim3 grid((out_dim + 4 * RPW - 1) / (4 * RPW)), block(256);
    const bf16 *tp = (const bf16*)t, *wp = (const bf16*)W, *bp = (const bf16*)bias;
    if (in_dim <= 512)
        silu_gemv_kernel<1, RPW><<<grid, block, 0, st>>>(tp, wp, bp, (bf16*)out, in_dim, out_dim, apply_silu);
    else if (in_dim <= 3072)
        silu_gemv_kernel<6, RPW><<<grid, block, 0, st>>>(tp, wp, bp, (bf16*)out, in_dim, out_dim, apply_silu);
    else
        silu_gemv_kernel<8, RPW><<<grid, block, 0, st>>>(tp, wp, bp, (bf16*)out, in_dim, out_dim, apply_silu);
    FP_CHECK_LAUNCH();
    return 0;
}

extern "C" int fp_rel_l1_acc_floats(void) { return 2 + 2 * REL_L1_MAX_BLOCKS; }

extern "C" int fp_rel_l1_bf16(const void* a, const void* b, int64_t n, float* acc, void* stream) {
    FP_REQUIRE(a && b && acc && n >= 0 && n % 8 == 0 && aligned16(a) && aligned16(b));
    hipStream_t st = (hipStream_t)stream;
    const int64_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < REL_L1_MAX_BLOCKS ? (n8 + 255) / 256 : REL_L1_MAX_BLOCKS);
    if (grid > 0) {
        rel_l1_kernel<<<grid, 256, 0, st>>>((const bf16*)a, (const bf16*)b, n8, acc + 2);
        FP_CHECK_LAUNCH();
    }
    rel_l1_final_kernel<<<1, 64, 0, st>>>(acc + 2, grid, acc);
    FP_CHECK_LAUNCH();
    return 0;
}
