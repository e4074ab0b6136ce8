// The MFMA GEMM pipeline shared by K5 (the DiT linears, bf16), K6 (the same linears in e4m3 on the fp8 MFMA) and
// K11 (the VAE convolutions and linears, fp16):
// C = epilogue(A . W^T + bias), 16-bit in / fp32 accumulate / 16-bit out  (gfx950 MFMA 16x16x32, wave64)
//
// The nn.Linear of every DiT block (ref hunyuan_video_packed.py:119-121, 133-135, 154-156, 167-169, 552,
// 570 and the diffusers FeedForward it calls at :646-647), with the elementwise work that follows each of
// them in the reference fused into the epilogue: bias, tanh-GELU (optionally only on a column range: the
// fused [k|v|q|mlp] projection of the single-stream blocks) and the gated residual `R + gate * (.)`.
//
// Both operands are K-contiguous ("NT"): A [M,K] tokens x features, W [N,K] nn.Linear layout.
//
// Structure (one workgroup = 8 waves = one 256 x 256 output tile, K step 64, 1 workgroup per CU):
//   * Operand tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction),
//     two 64 KiB buffers (A tile | W tile).  The LDS image is [row][128 B] with the 16-B chunk index XORed
//     with (row >> 1) & 7; the DMA writes linearly, so the XOR is applied to each lane's SOURCE address and
//     again on the fragment read: the ds_read_b128 fragment reads are bank-conflict free and every DMA
//     instruction still reads 8 whole 128-B lines.
//   * Each K tile is four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each (16 MFMAs), and
//     the operand tiles are staged as four "half tiles" named after the quadrant rows/columns they feed
//     (A-h0/A-h1 = first/second 64 rows of every wave row, W-h0/W-h1 = first/second 32 columns of every
//     wave column).  A half tile is re-staged two phases after its last read, so the DMA of K tiles t+1 and
//     t+2 is always in flight under the MFMAs of tile t; one counted s_waitcnt vmcnt per K tile, never 0 in
//     the steady state.
//   * The two wave rows (waves 0-3 / 4-7, one of each per SIMD) run half a phase apart: while one does its
//     16 MFMAs the other issues its LDS reads and DMA, and they swap at every barrier.
//   * The MFMA is issued with the operands swapped (D^T = W.A^T): the accumulator then has the TOKEN ROW on
//     the lane and 4 consecutive output columns in consecutive registers, so the epilogue stores 8 B per lane
//     contiguous in N and bias/gate are per-register constants.
#pragma once
#include "fp_common.h"

namespace fp_gemm {

constexpr int BM = 256, BN = 256, BK = 64;   // BK: K step of the 16-bit types (fp8: 128); always 128 B per row
constexpr int OP_BYTES = BM * BK * 2;        // 32 KiB: one operand tile
constexpr int BUF_BYTES = 2 * OP_BYTES;      // A tile then W tile

// Geometry of the implicit-GEMM A operand of a causal 3x3x3 convolution over a channels-last volume
// (AMODE 1): output position m = (t, h, w) of a [To, Ho, Wo] grid, K index = tap * Cin + c with
// tap = (dt * 3 + dh) * 3 + dw.  Padding is REPLICATE (indices clamp), causal in time (two frames in front,
// none behind); the source may be a nearest-neighbour upsampling of a smaller [Ti, Hi, Wi] volume that is
// never materialised (st/sh/sw = log2 of the factor; in time the first frame is not doubled).
// st/sh/sw = 1: the input is nearest-upsampled x2 on that axis on the fly (decoder); dt/dh/dw = 1: the convolution
// has stride 2 on that axis (encoder downsamplers).  hmax/wmax: upper clamp of the tap coordinate (replicate
// padding) in the coordinate system the taps are applied in (upsampled input, or input).
// gm (gemm_w4.h, any A mode; 0 = 8): the tile walk takes gm row tiles x all column tiles at a time (gm < 0: -gm
// column tiles x all row tiles), so the 32 CUs of an XCD run gm A panels against 32 / gm W panels.  Measured at
// S = 19385 (tools/gemm_xp.py, FP_GEMM_VARIANT = 1 + 16 gm [+ 8 for column-major], one box per scan): against gm = 8
// the long-K GEMMs (out-projection, ff2) gain 3-6 % with 3 column tiles x all rows, the widest (in-projection) 3-5 %
// with gm = 2, and the K = 3072 ones of middling width (qkv, ff1) lose 2-3 % with anything but 8.
struct ConvGeom {
    int Ti, Hi, Wi, To, Ho, Wo, Cin, st, sh, sw, dt, dh, dw, hmax, wmax, gm;
    int lcin;      // log2(Cin) (gemm_w4.h's gather: Cin is a power of two there)
    // gemm_w4.h, convolutions: walk the output row tiles band by band - a band = `band` consecutive tiles of a frame,
    // taken through all To frames before the next band (`tf` = tiles per frame; 0 = frame after frame).  The three
    // temporal taps of a band then find their input frames in the caches instead of one frame (> 100 MB) later.
    int band, tf;
};

// OCP e4m3 storage type of the fp8 path (K6): one byte per element, so a 128-B LDS row holds 128 K values.
struct fp8e4 {
    unsigned char bits;
};
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// One lane's operand for one 16-row tile and one K tile is two 16-B LDS reads h[0], h[1].
//   16-bit types: two MFMA 16x16x32 (K = 32 each), one per read.
//   fp8:          ONE MFMA 16x16x128 (the block-scaled f8f6f4 instruction with unit block scales: twice the
//                 bf16 rate), whose 32-B operand is the two reads side by side.  Both operands are read with the
//                 same lane -> K mapping, so the product does not depend on the instruction's K order.
template <typename T> struct mfma_op;
template <> struct mfma_op<bf16> {
    static constexpr int KS = 2;
    static __device__ __forceinline__ f32x4 run(const i32x4 (&a)[2], const i32x4 (&b)[2], int ks, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[ks]),
                                                       __builtin_bit_cast(bf16x8, b[ks]), c, 0, 0, 0);
    }
};
template <> struct mfma_op<_Float16> {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    static constexpr int KS = 2;
    static __device__ __forceinline__ f32x4 run(const i32x4 (&a)[2], const i32x4 (&b)[2], int ks, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[ks]), __builtin_bit_cast(h8, b[ks]), c,
                                                      0, 0, 0);
    }
};
template <> struct mfma_op<fp8e4> {
    static constexpr int KS = 1;
    static __device__ __forceinline__ f32x4 run(const i32x4 (&a)[2], const i32x4 (&b)[2], int, f32x4 c) {
        const i32x8 av = __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 bv = __builtin_shufflevector(b[0], b[1], 0, 1, 2, 3, 4, 5, 6, 7);
        // cbsz = blgp = 0: both operands e4m3; scale operands 0 = no block scaling
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void dma16(const void* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
}

// tanh-GELU as v * sigmoid(2u), u = sqrt(2/pi) (v + 0.044715 v^3):  v / (1 + 2^(-2u log2 e)).
// Two transcendentals (v_exp_f32, v_rcp_f32) and four plain VALU per value; the approximate reciprocal (1 ulp) is far
// inside the 16-bit output rounding.
__device__ __forceinline__ float gelu_tanh(float v) {
    constexpr float k = 2.f * 0.7978845608028654f * 1.4426950408889634f;
    const float z = v * __builtin_fmaf(v * v, 0.044715f * k, k);
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z));
}

// EPI: 0 bias | 1 bias+GELU | 2 bias+GELU on columns >= gelu_from | 3 R + gate*(.) | 4 R + (.)
// AMODE: 0 = A is a row-major [M, K] matrix | 1 = A is gathered from a channels-last volume (ConvGeom)
// T = operand element type, TO = output / bias / residual / gate type.  a_scale [M] and w_scale [N] (fp32) are
// the dequantisation factors of the fp8 path: C = epilogue(a_scale[m] * w_scale[n] * (A . W^T) + bias).
template <typename T, typename TO, int EPI, int AMODE>
__global__ __launch_bounds__(512, 2) void gemm_pipe_kernel(const T* __restrict__ A, int64_t lda,
                                                           int64_t a2_delta, int k1,
                                                           const T* __restrict__ W, int64_t ldw,
                                                           const TO* __restrict__ bias, TO* __restrict__ C,
                                                           int64_t ldc, int M, int N, int K, int gelu_from,
                                                           const TO* __restrict__ R, int64_t ldr,
                                                           const TO* __restrict__ gate, int ntm, int ntn,
                                                           ConvGeom cg, const float* __restrict__ a_scale,
                                                           const float* __restrict__ w_scale) {
    typedef i32x4 frag_t;
    typedef TO Tx4 __attribute__((ext_vector_type(4)));
    constexpr bool SCALED = sizeof(T) == 1;
    constexpr int BKE = 128 / (int)sizeof(T);      // K elements per tile
    constexpr int CE = 16 / (int)sizeof(T);        // elements per 16-B chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [2][A 32 KiB | W 32 KiB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // tile order: XCD-contiguous, grouped 8 row-tiles tall so that co-running tiles share A and W panels
    constexpr int GM = 8;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GM * ntn;
    const int group = wid / per_group, in_group = wid % per_group;
    const int first_m = group * GM;
    const int gsz = min(ntm - first_m, GM);
    const int tm = first_m + in_group % gsz, tn = in_group / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA source pointers: operand x half x instruction.  One instruction = 8 tile rows x 128 B.
    const int prow = lane >> 3, pch = lane & 7;
    const T* pa[2][2];
    const T* pw[2][2];
    int arow0[2][2], wrow0[2][2];
    int ct[2][2], ch[2][2], cw[2][2], cch[2][2];     // AMODE 1: output position and source chunk of each DMA row
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            arow0[h][j] = j * 128 + h * 64 + wave * 8;
            const int p = wave + 8 * j;
            wrow0[h][j] = (p >> 2) * 64 + h * 32 + (p & 3) * 8;
            const int ar = arow0[h][j] + prow, wrw = wrow0[h][j] + prow;
            if (AMODE == 0) {
                pa[h][j] = A + (int64_t)min(m0 + ar, M - 1) * lda + (pch ^ ((ar >> 1) & 7)) * CE;
            } else {
                const int m = min(m0 + ar, M - 1);
                cw[h][j] = (m % cg.Wo) << cg.dw;            // centre tap in tap coordinates
                ch[h][j] = ((m / cg.Wo) % cg.Ho) << cg.dh;
                ct[h][j] = (m / (cg.Wo * cg.Ho)) << cg.dt;
                cch[h][j] = (pch ^ ((ar >> 1) & 7)) * CE;
            }
            pw[h][j] = W + (int64_t)min(n0 + wrw, N - 1) * ldw + (pch ^ ((wrw >> 1) & 7)) * CE;
        }
    auto issue_a = [&](int h, int t) {
        unsigned char* dst = smem + (t & 1) * BUF_BYTES;
        if (AMODE == 0) {
            const int64_t koff = (int64_t)t * BKE + ((t * BKE >= k1) ? a2_delta : 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16(pa[h][j] + koff, dst + arow0[h][j] * 128);
        } else {
            const int k0 = t * BKE;                 // one K tile lies inside one tap (Cin % 64 == 0)
            const int tap = k0 / cg.Cin, c0 = k0 - tap * cg.Cin;
            const int dt = tap / 9 - 2, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tu = max(ct[h][j] + dt, 0);
                const int hu = min(max(ch[h][j] + dh, 0), cg.hmax);
                const int wu = min(max(cw[h][j] + dw, 0), cg.wmax);
                const int ts = cg.st ? ((tu + 1) >> 1) : tu;
                const int hs = hu >> cg.sh, ws = wu >> cg.sw;
                const int64_t off = ((int64_t)(ts * cg.Hi + hs) * cg.Wi + ws) * cg.Cin + c0 + cch[h][j];
                dma16(A + off, dst + arow0[h][j] * 128);
            }
        }
    };
    auto issue_w = [&](int h, int t) {
        const int64_t koff = (int64_t)t * BKE;
        unsigned char* dst = smem + (t & 1) * BUF_BYTES + OP_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(pw[h][j] + koff, dst + wrow0[h][j] * 128);
    };

    // ---- fragment read offsets: lane reads tile row (lane & 15), swizzled chunks
    //   16-bit: chunk (lane >> 4) for ks = 0 and that + 4 for ks = 1   |   fp8: chunks 2 (lane >> 4) and + 1
    const int fr = lane & 15, fq = lane >> 4;
    constexpr unsigned SECOND = SCALED ? 16u : 64u;
    const unsigned off_k0 = (unsigned)(fr * 128 + (((SCALED ? 2 * fq : fq) ^ (fr >> 1)) << 4));
    const unsigned char* a_rd0 = smem + (wr * 128) * 128 + off_k0;
    const unsigned char* a_rd1 = smem + (wr * 128) * 128 + (off_k0 ^ SECOND);
    const unsigned char* w_rd0 = smem + OP_BYTES + (wc * 64) * 128 + off_k0;
    const unsigned char* w_rd1 = smem + OP_BYTES + (wc * 64) * 128 + (off_k0 ^ SECOND);

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag_t af[4][2], wf[2][2];

    const int nk = K / BKE;

    // ---- prologue: tile 0 whole, then the two half tiles of tile 1 that the loop does not issue itself
    issue_a(0, 0);
    issue_w(0, 0);
    issue_w(1, 0);
    issue_a(1, 0);
    if (nk > 1) {
        issue_a(0, 1);
        issue_w(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // second wave row runs one barrier behind

#define FP_LDS_A(BUF, MH)                                                                                     \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                        \
        const int o_ = (BUF) * BUF_BYTES + ((MH) * 64 + mi * 16) * 128;                                       \
        af[mi][0] = *reinterpret_cast<const frag_t*>(a_rd0 + o_);                                             \
        af[mi][1] = *reinterpret_cast<const frag_t*>(a_rd1 + o_);                                             \
    }
#define FP_LDS_W(BUF, NH)                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                        \
        const int o_ = (BUF) * BUF_BYTES + ((NH) * 32 + ni * 16) * 128;                                       \
        wf[ni][0] = *reinterpret_cast<const frag_t*>(w_rd0 + o_);                                             \
        wf[ni][1] = *reinterpret_cast<const frag_t*>(w_rd1 + o_);                                             \
    }
#define FP_MFMA_QUAD(MH, NH)                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < mfma_op<T>::KS; ++ks)                                             \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                      \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                  \
                acc[(MH) * 4 + mi][(NH) * 2 + ni] =                                                           \
                    mfma_op<T>::run(wf[ni], af[mi], ks, acc[(MH) * 4 + mi][(NH) * 2 + ni]);                   \
    __builtin_amdgcn_s_setprio(0);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    __builtin_amdgcn_s_barrier();                                                                             \
    __builtin_amdgcn_sched_barrier(0);

    // One K tile = four phases.  BUF is the LDS buffer of tile t (a literal so that addresses fold).
#define FP_KTILE(BUF, t)                                                                                      \
    {                                                                                                         \
        /* q0: quadrant (0,0) */                                                                              \
        FP_LDS_W(BUF, 0)                                                                                      \
        FP_LDS_A(BUF, 0)                                                                                      \
        if ((t) + 1 < nk) issue_a(1, (t) + 1);                                                                \
        FP_MFMA_QUAD(0, 0)                                                                                    \
        /* q1: quadrant (0,1) */                                                                              \
        FP_LDS_W(BUF, 1)                                                                                      \
        if ((t) + 1 < nk) issue_w(0, (t) + 1);                                                                \
        FP_MFMA_QUAD(0, 1)                                                                                    \
        /* q2: quadrant (1,1) */                                                                              \
        FP_LDS_A(BUF, 1)                                                                                      \
        if ((t) + 2 < nk) issue_a(0, (t) + 2);                                                                \
        FP_MFMA_QUAD(1, 1)                                                                                    \
        /* q3: quadrant (1,0); the wait retires tile t+1 for the next phase */                                \
        FP_LDS_W(BUF, 0)                                                                                      \
        if ((t) + 2 < nk) {                                                                                   \
            issue_w(1, (t) + 2);                                                                              \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                  \
        } else {                                                                                              \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
        }                                                                                                     \
        FP_MFMA_QUAD(1, 0)                                                                                    \
    }

    int t = 0;
    for (; t + 1 < nk; t += 2) {
        FP_KTILE(0, t)
        FP_KTILE(1, t + 1)
    }
    if (t < nk) FP_KTILE(0, t)
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef FP_KTILE
#undef FP_MFMA_QUAD
#undef FP_LDS_W
#undef FP_LDS_A

    // ---- epilogue: lane (fr,fq) holds row m = m0 + wr*128 + mi*16 + fr, columns n0 + wc*64 + ni*16 + 4 fq + (0..3)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc * 64 + ni * 16 + 4 * fq;
        if (n >= N) continue;
        Tx4 bv = {0, 0, 0, 0};
        if (bias) bv = *reinterpret_cast<const Tx4*>(bias + n);
        Tx4 gv = {0, 0, 0, 0};
        if (EPI == 3) gv = *reinterpret_cast<const Tx4*>(gate + n);
        const bool do_gelu = EPI == 1 || (EPI == 2 && n >= gelu_from);
        f32x4 ws = {1.f, 1.f, 1.f, 1.f};
        if (SCALED) ws = *reinterpret_cast<const f32x4*>(w_scale + n);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wr * 128 + mi * 16 + fr;
            if (m >= M) continue;
            float v[4];
            if (SCALED) {
                const float as = a_scale[m];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][j] * (as * ws[j]) + (float)bv[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][j] + (float)bv[j];
            }
            if (do_gelu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_tanh(v[j]);
            }
            if (EPI == 3 || EPI == 4) {
                const Tx4 rv = *reinterpret_cast<const Tx4*>(R + (int64_t)m * ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (float)rv[j] + (EPI == 3 ? (float)gv[j] : 1.f) * v[j];
            }
            Tx4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (TO)v[j];
            *reinterpret_cast<Tx4*>(C + (int64_t)m * ldc + n) = o;
        }
    }
}


template <typename T, int EPI, int AMODE, typename TO = T>
int launch(const void* a, int64_t lda, int64_t a2_delta, int k1, const void* w, int64_t ldw, const void* bias,
           void* c, int64_t ldc, int M, int N, int K, int gelu_from, const void* resid, int64_t ldr,
           const void* gate, const ConvGeom& cg, hipStream_t st, const float* a_scale = nullptr,
           const float* w_scale = nullptr) {
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    constexpr int smem = 2 * BUF_BYTES;
    auto kern = gemm_pipe_kernel<T, TO, EPI, AMODE>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    kern<<<dim3((unsigned)(ntm * ntn)), dim3(512), smem, st>>>((const T*)a, lda, a2_delta, k1, (const T*)w, ldw,
                                                              (const TO*)bias, (TO*)c, ldc, M, N, K, gelu_from,
                                                              (const TO*)resid, ldr, (const TO*)gate, ntm, ntn, cg,
                                                              a_scale, w_scale);
    FP_CHECK_LAUNCH();
    return 0;
}

// Argument checks shared by the dense entry points (element size 2 for both dtypes).
inline int check_dense(const void* a, int64_t lda, const void* a2, int& k1, const void* w, int64_t ldw,
                       const void* bias, const void* c, int64_t ldc, int M, int N, int K, int epilogue,
                       int gelu_from, const void* resid, int64_t ldr, const void* gate, int64_t& a2_delta) {
    FP_REQUIRE(a && w && c && M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0);
    FP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && ldw >= K && ldc >= N);
    FP_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 7) == 0);
    FP_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0);
    a2_delta = 0;
    if (a2) {
        FP_REQUIRE(k1 > 0 && k1 < K && k1 % BK == 0 && ((uintptr_t)a2 & 15) == 0 && lda >= k1 && lda >= K - k1);
        // element offset that turns "A + k" into "A2 + (k - k1)" for k >= k1 (same row stride)
        a2_delta = ((const char*)a2 - (const char*)a) / 2 - k1;
    } else {
        FP_REQUIRE(lda >= K);
        k1 = K;
    }
    FP_REQUIRE(epilogue >= 0 && epilogue <= 4);
    if (epilogue == 2) FP_REQUIRE(gelu_from >= 0 && gelu_from % 4 == 0);
    if (epilogue >= 3) FP_REQUIRE(resid && ldr % 4 == 0 && ldr >= N && ((uintptr_t)resid & 7) == 0);
    if (epilogue == 3) FP_REQUIRE(gate && ((uintptr_t)gate & 7) == 0);
    FP_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31));
    return 0;
}

}  // namespace fp_gemm
