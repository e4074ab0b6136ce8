// K11: kernels of the causal-3D VAE decoder (fp16, channels-last [T, H, W, C] volumes).
//
// Replaces what ref diffusers_helper/hunyuan.py:93-103 (vae_decode) reaches inside diffusers 0.33.1's
// AutoencoderKLHunyuanVideo.decode (third-party, not in the reference tree; restated in oracle/vae_oracle.py):
//   * HunyuanVideoCausalConv3d 3x3x3 (replicate padding, two frames in front and none behind), optionally
//     reading its input through the nearest-neighbour upsampling of HunyuanVideoUpsampleCausal3D without
//     materialising it, with bias and the resnet's residual add in the epilogue  -> implicit GEMM on the
//     MFMA pipeline of gemm_pipeline.h (the A operand is gathered by LDS-DMA, one tap per K tile);
//   * 1x1x1 convolutions and the mid-block attention's linears / QK^T / PV                -> dense fp16 GEMM;
//   * GroupNorm(32) (+SiLU): deterministic two-stage statistics, then one fused normalise+affine+SiLU pass;
//   * the mid-block attention (one head of 512 channels): fp32 logits of a chunk of query rows (the dense GEMM with
//     an fp32 epilogue), frame-causal softmax into fp16 in place, P.V - the S x S matrix is never materialised.
#include "gemm_w4.h"
int fpi_gemm_variant();      // gemm_bf16.hip: the diagnostics switch (bit 2: no tall convolution tile)

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

extern "C" int fp_gemm_f16(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw,
                           const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue,
                           int gelu_from, const void* resid, int64_t ldr, const void* gate, void* stream) {
    using namespace fp_gemm;
    int64_t a2_delta;
    if (int rc = check_dense(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, epilogue, gelu_from, resid, ldr, gate, a2_delta))
        return rc;
    if (M == 0) return 0;
    const ConvGeom cg{};
    hipStream_t st = (hipStream_t)stream;
    // 16-byte aligned output (and residual) rows go to the 4-wave kernel (gemm_w4.h), anything else to the 8-wave one
    const bool wide = ((uintptr_t)c & 15) == 0 && ldc % 8 == 0 &&
                      (epilogue < 3 || (((uintptr_t)resid & 15) == 0 && ldr % 8 == 0));
    switch (epilogue) {
        case 0:
            if (wide && N <= 128) return launch_w4<f16, 0, 0, f16, 4>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st);
            return wide ? launch_w4<f16, 0, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st)
                        : launch<f16, 0, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st);
        case 4:
            if (wide && N <= 128) return launch_w4<f16, 4, 0, f16, 4>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st);
            return wide ? launch_w4<f16, 4, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st)
                        : launch<f16, 4, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st);
        default: return (int)hipErrorInvalidValue;      // the VAE uses bias-only and bias+residual
    }
}

// c [M, N] fp32 = a [M, K] . w [N, K]^T, fp16 operands, no bias: the mid-block attention's logits (kept in fp32: one
// head of 512 channels, and an fp16 logit carries an absolute error of |s| 2^-11 into the exponent).
extern "C" int fp_gemm_f16_f32out(const void* a, int64_t lda, const void* w, int64_t ldw, void* c, int64_t ldc, int M,
                                  int N, int K, void* stream) {
    using namespace fp_gemm;
    FP_REQUIRE(a && w && c && M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0);
    FP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && lda >= K && ldw >= K && ldc >= N);
    FP_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 15) == 0);
    FP_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31));
    if (M == 0) return 0;
    const ConvGeom cg{};
    return launch_w4<f16, 0, 0, float>(a, lda, 0, K, w, ldw, nullptr, c, ldc, M, N, K, 0, nullptr, 0, nullptr, cg,
                                       (hipStream_t)stream);
}

// Band-major tile walk for a convolution over more than one frame (ConvGeom::band): bands of >= 16 row tiles (4 096
// output voxels: about 1 MiB of a 128-channel input frame, three frames of it inside an XCD's 4 MiB L2) when a frame is
// a whole number of tiles.  FRAMEPACK_CONV_BAND=0 keeps the frame-after-frame walk (diagnostics, read once).
static void pick_band(fp_gemm::ConvGeom& cg, int tile_rows) {
    static const int mode = [] { const char* e = getenv("FRAMEPACK_CONV_BAND"); return e ? atoi(e) : -1; }();
    cg.band = cg.tf = 0;
    const int64_t frame = (int64_t)cg.Ho * cg.Wo;
    if (mode == 0 || cg.To < 2 || frame % tile_rows != 0) return;
    const int tf = (int)(frame / tile_rows);
    const int want = (mode > 0 ? mode : 16) * fp_gemm::BM / tile_rows;      // (a band = 4 096 output voxels or more)
    int band = 0;
    for (int b = want; b <= tf; ++b)
        if (tf % b == 0) { band = b; break; }
    if (band == 0 || band == tf) return;
    cg.band = band;
    cg.tf = tf;
}

static int conv_launch(const void* x, const void* w, const void* bias, void* y, const fp_gemm::ConvGeom& cg_in, int Cout,
                       const void* resid, hipStream_t st) {
    using namespace fp_gemm;
    ConvGeom cg = cg_in;
    const int64_t M64 = (int64_t)cg.To * cg.Ho * cg.Wo;
    FP_REQUIRE(M64 < (1ll << 31) && (int64_t)cg.Ti * cg.Hi * cg.Wi * cg.Cin < (1ll << 40));
    // centre taps are packed t << 22 | h << 11 | w in gemm_w4.h, in tap coordinates (strided: twice the output's)
    FP_REQUIRE((cg.To << cg.dt) < 1024 && (cg.Ho << cg.dh) < 2048 && (cg.Wo << cg.dw) < 2048);
    const int M = (int)M64, N = Cout, K = 27 * cg.Cin;
    FP_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + 31) / 32) < (1ll << 31));
    // the 4-wave kernel's gather needs Cin to be a power of two (voxel steps become shifts) of at least two K tiles
    // (128 channels: K tiles 2 i and 2 i + 1 then share their tap, and the odd one reuses the even one's source offsets),
    // 24-bit multiplies of the plane and row pitch in voxels, and every source of a tile within 4 GiB of the tile's base
    // (the start of the frame its earliest tap reads: a tile's 512 rows span 512 / (Ho Wo) + 2 output frames at most,
    // twice as many input frames under a temporal stride, and the taps reach two frames back); anything else (and
    // outputs aligned to 8 bytes only) runs on the 8-wave kernel, which computes every source from coordinates
    const int64_t tile_frames = 2 * (512 / ((int64_t)cg.Ho * cg.Wo) + 2) + 4;
    const bool pow2 = (cg.Cin & (cg.Cin - 1)) == 0 && cg.Cin >= 2 * BK && cg.lcin == __builtin_ctz((unsigned)cg.Cin) &&
                      (int64_t)cg.Hi * cg.Wi < (1 << 23) && (int64_t)cg.Ti * cg.Hi * cg.Wi < (1ll << 31) &&
                      tile_frames * cg.Hi * cg.Wi * cg.Cin * 2 < (1ll << 32);
    const bool wide = pow2 && ((uintptr_t)y & 15) == 0 && (!resid || ((uintptr_t)resid & 15) == 0);      // Cout % 8 == 0
    // Cout = 128 over at least two rounds of 512-row tiles (the last up block at full resolution): the tall tile of
    // gemm_w4.h (512 x 128).  fp_debug_gemm_variant(1 | 4) keeps the 256 x 128 tile (diagnostics: A/B, tests).
    const bool tall_on = (fpi_gemm_variant() & 4) == 0;
    if (wide && tall_on && N == 128 && M64 >= 2ll * 512 * persistent_grid()) {
        pick_band(cg, 2 * BM);
        return resid ? launch_w4<f16, 4, 1, f16, 8, 0, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, resid, Cout, nullptr, cg, st)
                     : launch_w4<f16, 0, 1, f16, 8, 0, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, nullptr, 0, nullptr, cg, st);
    }
    pick_band(cg, BM);
    // Cout <= 128 / <= 32 (the last up block and conv_out): the narrow tiles of gemm_w4.h
    if (wide && N <= 32 && !resid)
        return launch_w4<f16, 0, 1, f16, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, nullptr, 0, nullptr, cg, st);
    if (wide && N <= 128)
        return resid ? launch_w4<f16, 4, 1, f16, 4>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, resid, Cout, nullptr, cg, st)
                     : launch_w4<f16, 0, 1, f16, 4>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, nullptr, 0, nullptr, cg, st);
    if (resid)
        return wide ? launch_w4<f16, 4, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, resid, Cout, nullptr, cg, st)
                    : launch<f16, 4, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, resid, Cout, nullptr, cg, st);
    return wide ? launch_w4<f16, 0, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, nullptr, 0, nullptr, cg, st)
                : launch<f16, 0, 1>(x, 0, 0, K, w, K, bias, y, Cout, M, N, K, 0, nullptr, 0, nullptr, cg, st);
}

// y[To,Ho,Wo,Cout] = conv3x3x3_causal_replicate(upsample(x[Ti,Hi,Wi,Cin])) + bias (+ resid)
// w: [Cout][27][Cin] (tap-major, channel-minor);  up_t/up_h/up_w in {1, 2};  To = up_t == 2 ? 2 Ti - 1 : Ti.
extern "C" int fp_conv3d_causal_f16(const void* x, const void* w, const void* bias, void* y, int Ti, int Hi, int Wi,
                                    int up_t, int up_h, int up_w, int Cin, int Cout, const void* resid,
                                    void* stream) {
    using namespace fp_gemm;
    FP_REQUIRE(x && w && y && Ti > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cin % BK == 0 && Cout > 0 && Cout % 8 == 0);
    FP_REQUIRE((up_t == 1 || up_t == 2) && (up_h == 1 || up_h == 2) && (up_w == 1 || up_w == 2));
    FP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 7) == 0);
    FP_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0);
    FP_REQUIRE(!resid || ((uintptr_t)resid & 7) == 0);
    ConvGeom cg{};
    cg.Ti = Ti; cg.Hi = Hi; cg.Wi = Wi;
    cg.To = up_t == 2 ? 2 * Ti - 1 : Ti; cg.Ho = Hi * up_h; cg.Wo = Wi * up_w;
    cg.Cin = Cin; cg.st = up_t - 1; cg.sh = up_h - 1; cg.sw = up_w - 1;
    cg.hmax = cg.Ho - 1; cg.wmax = cg.Wo - 1;
    cg.lcin = __builtin_ctz((unsigned)Cin);
    return conv_launch(x, w, bias, y, cg, Cout, resid, (hipStream_t)stream);
}

// Strided form (the encoder's downsamplers): y[To,Ho,Wo,Cout] = conv3x3x3 with stride (dn_t, dn_h, dn_w), each 1 or 2,
// over x padded causally in time (two frames in front) and by one voxel on every spatial side, replicate mode:
// To = (Ti - 1) / dn_t + 1, Ho = (Hi - 1) / dn_h + 1, Wo = (Wi - 1) / dn_w + 1.
extern "C" int fp_conv3d_causal_strided_f16(const void* x, const void* w, const void* bias, void* y, int Ti, int Hi,
                                            int Wi, int dn_t, int dn_h, int dn_w, int Cin, int Cout, void* stream) {
    using namespace fp_gemm;
    FP_REQUIRE(x && w && y && Ti > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cin % BK == 0 && Cout > 0 && Cout % 8 == 0);
    FP_REQUIRE((dn_t == 1 || dn_t == 2) && (dn_h == 1 || dn_h == 2) && (dn_w == 1 || dn_w == 2));
    FP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 7) == 0);
    FP_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0);
    ConvGeom cg{};
    cg.Ti = Ti; cg.Hi = Hi; cg.Wi = Wi;
    cg.To = (Ti - 1) / dn_t + 1; cg.Ho = (Hi - 1) / dn_h + 1; cg.Wo = (Wi - 1) / dn_w + 1;
    cg.Cin = Cin; cg.dt = dn_t - 1; cg.dh = dn_h - 1; cg.dw = dn_w - 1;
    cg.hmax = Hi - 1; cg.wmax = Wi - 1;
    cg.lcin = __builtin_ctz((unsigned)Cin);
    return conv_launch(x, w, bias, y, cg, Cout, nullptr, (hipStream_t)stream);
}

namespace {

constexpr int GN_THREADS = 256;

// Stage 1 of the GroupNorm statistics: block b sums x and x^2 of its row range per 4-channel quad.
// x: [rows, C] fp16, C % 8 == 0, C / 8 <= 256 and a power of two.  partial: [gridDim.x][C / 4][2] floats.
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const f16* __restrict__ x, int64_t rows, int C,
                                                                float* __restrict__ partial) {
    __shared__ float red[GN_THREADS * 4];
    const int chunks = C >> 3;                       // 16-B chunks per row
    const int rpi = GN_THREADS / chunks;             // rows per iteration of the block
    const int c8 = threadIdx.x % chunks, rl = threadIdx.x / chunks;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * rpi + rl; r < rows; r += (int64_t)gridDim.x * rpi) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(x + r * C + c8 * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = (float)v[j], b = (float)v[4 + j];
            s0 += a; q0 += a * a; s1 += b; q1 += b * b;
        }
    }
    red[threadIdx.x * 4 + 0] = s0; red[threadIdx.x * 4 + 1] = q0;
    red[threadIdx.x * 4 + 2] = s1; red[threadIdx.x * 4 + 3] = q1;
    __syncthreads();
    if (rl == 0) {                                   // fixed-order sum over the block's row lanes
        for (int i = 1; i < rpi; ++i) {
            const float* p = red + (i * chunks + c8) * 4;
            s0 += p[0]; q0 += p[1]; s1 += p[2]; q1 += p[3];
        }
        float* o = partial + ((int64_t)blockIdx.x * (C >> 2) + c8 * 2) * 2;
        o[0] = s0; o[1] = q0; o[2] = s1; o[3] = q1;
    }
}

// Stage 2: stats[g] = (mean, rstd) of group g (C / groups channels).  One wave per group: lane l sums the
// partials of blocks l, l+64, ... in order, then the lanes are combined by a fixed butterfly: deterministic.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                         int groups, int64_t rows, float eps,
                                                         float* __restrict__ stats) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int quads = (C / groups) >> 2;             // 4-channel quads per group
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblocks; b += 64)
        for (int i = 0; i < quads; ++i) {
            const float* p = partial + ((int64_t)b * (C >> 2) + g * quads + i) * 2;
            s += p[0]; q += p[1];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) {
        const double n = (double)rows * (C / groups);
        const double mean = s / n;
        const double var = fmax(q / n - mean * mean, 0.0);
        stats[2 * g] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// y = f((x - mean_g) * rstd_g * gamma_c + beta_c),  f = SiLU or identity.
// The grid's stride (gridDim.x * 256 chunks) is a multiple of the chunks per row (C / 8, a power of two <= 256), so a
// thread keeps its 8 channels for all its rows: their scale a = rstd * gamma and offset b = beta - mean * a are
// computed once, and a row costs one 16-byte load, 8 fma (+ SiLU: exp2, add, rcp, mul) and one 16-byte store.
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t n8,
                                                       int C, int groups, const float* __restrict__ stats,
                                                       const f16* __restrict__ gamma, const f16* __restrict__ beta) {
    const int chunks = C >> 3, gs = C / groups;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = (int)(i0 % chunks) * 8;
    const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + c0);
    const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + c0);
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / gs;
        a[j] = stats[2 * g + 1] * (float)gm[j];
        b[j] = (float)bt[j] - stats[2 * g] * a[j];
    }
    for (int64_t i = i0; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(x + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = __builtin_fmaf((float)v[j], a[j], b[j]);
            if (SILU) t = t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
            o[j] = (f16)t;
        }
        *reinterpret_cast<f16x8*>(y + i * 8) = o;
    }
}

// Frame-causal softmax of fp32 logits, the fp16 probabilities written IN PLACE over the front half of each row
// (row i: floats s[i * ld ..], halves ((f16*)(s + i * ld))[0 .. cols)).  Row i (query row0 + i) keeps the columns below
// valid = min(((row0 + i) / q_per_frame + 1) * k_per_frame, cols); the columns [valid, cols) become 0.
// One workgroup per row.  Pass 1: max and sum.  Pass 2, in chunks of 2048 columns: all threads read their 8 floats,
// barrier, write their 8 halves - the halves of a chunk land on floats of this or an earlier chunk, all read by then.
__global__ __launch_bounds__(256) void softmax_f32_to_f16_kernel(float* __restrict__ s, int64_t ld, int cols, int row0,
                                                                 int q_per_frame, int k_per_frame, float scale_log2e) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    float* p = s + (int64_t)row * ld;
    f16* ph = reinterpret_cast<f16*>(p);
    const int64_t lim = ((int64_t)(row0 + row) / q_per_frame + 1) * k_per_frame;
    const int valid = (int)(lim < cols ? lim : cols);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY, l = 0.f;
    for (int c = threadIdx.x * 8; c < valid; c += 256 * 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + c), b = *reinterpret_cast<const f32x4*>(p + c + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        float mx = v[0];                                               // column c < valid
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, c + j < valid ? v[j] : -INFINITY);
        const float mn = fmaxf(m, mx * scale_log2e);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += c + j < valid ? __builtin_amdgcn_exp2f(v[j] * scale_log2e - mn) : 0.f;
        l = l * __builtin_amdgcn_exp2f(m - mn) + acc;
        m = mn;
    }
    float mw = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    if (lane == 0) red[wave] = mw;
    __syncthreads();
    const float mt = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lw = (m == -INFINITY) ? 0.f : l * __builtin_amdgcn_exp2f(m - mt);
    lw = wave_sum(lw);
    __syncthreads();
    if (lane == 0) red[4 + wave] = lw;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int base = 0; base < valid; base += 256 * 8) {               // (uniform trip count)
        const int c = base + threadIdx.x * 8;
        f16x8 o;
        if (c < valid) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + c), b = *reinterpret_cast<const f32x4*>(p + c + 4);
            const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = c + j < valid ? (f16)(__builtin_amdgcn_exp2f(v[j] * scale_log2e - mt) * inv) : (f16)0.f;
        }
        __syncthreads();
        if (c < valid) *reinterpret_cast<f16x8*>(ph + c) = o;
    }
    __syncthreads();                                                   // every logit has been read: the zero tail may land
    for (int c = (valid + 7) / 8 * 8 + threadIdx.x * 8; c < cols; c += 256 * 8) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)0.f;
        *reinterpret_cast<f16x8*>(ph + c) = o;
    }
}

}  // namespace

// GroupNorm statistics of x [rows, C] (channels-last volume flattened): stats[g] = (mean, rstd), fp32.
// workspace: >= fp_groupnorm_workspace_floats(C) floats of device memory (fully overwritten by the call).
extern "C" int fp_groupnorm_workspace_floats(int C) { return 1024 * (C / 4) * 2; }

extern "C" int fp_groupnorm_stats_f16(const void* x, int64_t rows, int C, int groups, float eps, float* workspace,
                                      float* stats, void* stream) {
    FP_REQUIRE(x && workspace && stats && rows > 0 && C > 0 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0);
    FP_REQUIRE(C % 8 == 0 && (C / 8) <= GN_THREADS && ((C / 8) & (C / 8 - 1)) == 0 && ((uintptr_t)x & 15) == 0);
    const int rpi = GN_THREADS / (C / 8);
    const int nblocks = (int)((rows + rpi - 1) / rpi < 1024 ? (rows + rpi - 1) / rpi : 1024);
    hipStream_t st = (hipStream_t)stream;
    gn_partial_kernel<<<dim3(nblocks), dim3(GN_THREADS), 0, st>>>((const f16*)x, rows, C, workspace);
    FP_CHECK_LAUNCH();
    gn_finalize_kernel<<<dim3(groups), dim3(64), 0, st>>>(workspace, nblocks, C, groups, rows, eps, stats);
    FP_CHECK_LAUNCH();
    return 0;
}

// y = f(GroupNorm(x; stats) * gamma + beta), f = SiLU (silu != 0) or identity; y may alias x.
extern "C" int fp_groupnorm_apply_f16(const void* x, void* y, int64_t rows, int C, int groups, const float* stats,
                                      const void* gamma, const void* beta, int silu, void* stream) {
    FP_REQUIRE(x && y && stats && gamma && beta && rows > 0 && C % 8 == 0 && groups > 0 && C % groups == 0);
    FP_REQUIRE((C / 8) <= 256 && ((C / 8) & (C / 8 - 1)) == 0);        // a thread keeps its channels (kernel comment)
    FP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 &&
               ((uintptr_t)beta & 15) == 0);
    const int64_t n8 = rows * (C / 8);
    const int nblocks = (int)(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (silu)
        gn_apply_kernel<true><<<dim3(nblocks), dim3(256), 0, st>>>((const f16*)x, (f16*)y, n8, C, groups, stats,
                                                                   (const f16*)gamma, (const f16*)beta);
    else
        gn_apply_kernel<false><<<dim3(nblocks), dim3(256), 0, st>>>((const f16*)x, (f16*)y, n8, C, groups, stats,
                                                                    (const f16*)gamma, (const f16*)beta);
    FP_CHECK_LAUNCH();
    return 0;
}

// Frame-causal softmax of the mid-block attention's fp32 logits s [rows, cols] (row stride ld floats); the fp16
// probabilities overwrite the front of each row: P = (f16*)s viewed [rows, cols] with row stride 2 ld halves.
// Query i of the call is token row0 + i; it attends to the keys below ((row0 + i) / q_per_frame + 1) * k_per_frame.
extern "C" int fp_softmax_framecausal_f32_f16(void* s, int rows, int cols, int64_t ld, int row0, int q_per_frame,
                                              int k_per_frame, float scale, void* stream) {
    FP_REQUIRE(s && rows >= 0 && cols > 0 && cols % 8 == 0 && ld >= cols && ld % 4 == 0 && row0 >= 0);
    FP_REQUIRE(q_per_frame > 0 && k_per_frame > 0 && ((uintptr_t)s & 15) == 0);
    if (rows == 0) return 0;
    softmax_f32_to_f16_kernel<<<dim3(rows), dim3(256), 0, (hipStream_t)stream>>>((float*)s, ld, cols, row0, q_per_frame,
                                                                                 k_per_frame, scale * 1.4426950408889634f);
    FP_CHECK_LAUNCH();
    return 0;
}
