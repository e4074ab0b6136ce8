// Second-generation GEMM main loop: ONE wave per SIMD (4 waves = one 256 x 256 output tile), each wave owning a
// 128 x 128 block whose 256 accumulator registers live in the AGPR half of the register file.
//
// Why (measured, profiles/r1_e_gemm_pmc.md): the 8-wave ping-pong pipeline of gemm_pipeline.h keeps the matrix pipe
// 72 % busy (two barriers per 16 MFMAs); the rest is hand-off.  With one wave per SIMD there is nothing to hand
// off: the wave interleaves its own LDS reads and LDS-DMA issues between MFMAs and meets the other three waves
// only twice per K tile (every 64 MFMAs = 1024 matrix-pipe cycles).
//
// Used by: fp_gemm_bf16 / fp_gemm_f16 / fp_conv3d_causal*_f16 (16-bit MFMA 16x16x32, two K steps per tile: described
// below) and fp_gemm_fp8_bf16 (e4m3, ONE MFMA 16x16x128 per tile and K tile: the `KS == 1` branch of `phase`).
// Template NT narrows the tile for narrow outputs (see the kernel's comment); ConvGeom::gm sets the tile walk.
//
// Same LDS image, swizzle, operand swap (token row on the lane) and epilogues as gemm_pipeline.h.
//   * K tile = 64 (128 B per row), two buffers of 32 KiB per operand, staged as THREE units per tile: A-h0 / A-h1 (rows 0-63 /
//     64-127 of each wave row) and W (whole).  A K tile is two phases: phase p multiplies A-h{p} (4 row tiles)
//     with all of W (8 column tiles, fragments kept in registers across both phases) = 64 MFMAs.
//   * A unit is re-staged right after the barrier that follows its last read: during P0(t) the wave issues
//     A-h1(t+1), during P1(t) A-h0(t+2) and W(t+2); every unit is in flight for >= 2 phases, and the waits are
//     counted (vmcnt 16 before every phase), never 0.  Loads past the last tile are clamped to it (they land
//     in a buffer nobody reads), so the loop body has no branch and stays one scheduling region.
//   * The last two MFMA groups of a phase are issued AFTER the next barrier, behind the first LDS reads of the
//     new phase, so the read latency after a barrier is covered by 256 cycles of matrix work.
#pragma once
#include <type_traits>
#include "gemm_pipeline.h"

namespace fp_gemm {

template <int N> using ic = std::integral_constant<int, N>;
template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic<I>{});
        static_for<N, I + 1>(f);
    }
}

// Makes the kernel descriptor reserve all 256 AGPRs (the asm below addresses them by number).
__device__ __forceinline__ void reserve_agprs() {
    asm volatile("" ::: "a0", "a255");
}

// a[IDX:IDX+3] (+)= w . a   (operands swapped as in gemm_pipeline.h: the token row ends up on the lane)
// LGKM >= 0: at most LGKM LDS reads may still be in flight when the MFMA issues (the 16-bit main loop reads its
// fragments with inline asm, invisible to the compiler's wait-count bookkeeping - see rd_a)
template <typename T, int IDX, bool ZERO_C, int LGKM = -1>
__device__ __forceinline__ void mfma_acc(const i32x4& w, const i32x4& a) {
    if constexpr (LGKM >= 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LGKM));
    if constexpr (std::is_same<T, bf16>::value) {
        if constexpr (ZERO_C)
            asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, 0" ::"n"(IDX), "n"(IDX + 3), "v"(w), "v"(a));
        else
            asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(IDX), "n"(IDX + 3), "v"(w), "v"(a));
    } else {
        if constexpr (ZERO_C)
            asm volatile("v_mfma_f32_16x16x32_f16 a[%0:%1], %2, %3, 0" ::"n"(IDX), "n"(IDX + 3), "v"(w), "v"(a));
        else
            asm volatile("v_mfma_f32_16x16x32_f16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(IDX), "n"(IDX + 3), "v"(w), "v"(a));
    }
}
// a[IDX:IDX+3] (+)= w . a for e4m3 operands: ONE 16x16x128 MFMA (32 bytes per lane and operand) per tile and K tile
template <int IDX, bool ZERO_C>
__device__ __forceinline__ void mfma8_acc(const i32x8& w, const i32x8& a) {
    if constexpr (ZERO_C)
        asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%0:%1], %2, %3, 0" ::"n"(IDX), "n"(IDX + 3), "v"(w), "v"(a));
    else
        asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(IDX), "n"(IDX + 3), "v"(w), "v"(a));
}
template <int IDX> __device__ __forceinline__ f32x4 acc_read() {
    float r0, r1, r2, r3;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\nv_accvgpr_read_b32 %1, a[%5]\nv_accvgpr_read_b32 %2, a[%6]\n"
                 "v_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                 : "n"(IDX), "n"(IDX + 1), "n"(IDX + 2), "n"(IDX + 3));
    return f32x4{r0, r1, r2, r3};
}

template <int OFF> __device__ __forceinline__ void lds_read_b128(i32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// LDS reads that may still be in flight when group g (0..5) of phase P issues its first MFMA, from the order in which
// the 16-bit loop issues them (LDS reads return in order).  P0: [W(ks 0, ni < NT), A0, A1] behind the barrier, then
// behind group j's MFMAs 0 / 1 / 2: A(j+2), W(ks 1, 2j), W(ks 1, 2j+1) (the W reads for j < 4 and ni < NT only: a
// fragment no MFMA uses is never read - its register would be handed out again while the read is in flight).
// Group g needs A(g) and W(ks g/4).  P1: [A0, A1], then A(j+2) behind group j's first MFMA; W is in registers.
#ifndef FP_A_AHEAD
#define FP_A_AHEAD 2
#endif
constexpr int A_AHEAD = FP_A_AHEAD;             // an A fragment is read this many groups (of 8 MFMAs) before its own
constexpr int A_RING = A_AHEAD <= 2 ? 4 : 8;
constexpr int lgkm_before(int P, int g, int NT) {
    int issued = 0, idx_a[16] = {}, last_w1 = -1;
    if (P == 0) issued += NT < 8 ? NT : 8;      // W(ks 0)
    for (int i = 0; i < A_AHEAD; ++i) idx_a[i] = issued++;
    for (int j = 0; j < g; ++j) {               // what groups 0 .. g-1 issue
        if (j + A_AHEAD < 8) idx_a[j + A_AHEAD] = issued++;
        if (P == 0 && j < 4) {
            if (2 * j < NT) last_w1 = issued++;
            if (2 * j + 1 < NT) last_w1 = issued++;
        }
    }
    int need = idx_a[g];
    if (P == 0 && g >= 4 && last_w1 > need) need = last_w1;
    return issued - (need + 1);
}
#if FP_A_AHEAD == 2
static_assert(lgkm_before(0, 0, 8) == 1 && lgkm_before(0, 1, 8) == 3 && lgkm_before(0, 2, 8) == 5 &&
              lgkm_before(0, 3, 8) == 5 && lgkm_before(0, 4, 8) == 0 && lgkm_before(0, 5, 8) == 1, "wait counts, NT = 8");
static_assert(lgkm_before(0, 0, 4) == 1 && lgkm_before(0, 1, 4) == 3 && lgkm_before(0, 2, 4) == 5 &&
              lgkm_before(0, 3, 4) == 3 && lgkm_before(0, 4, 4) == 1 && lgkm_before(0, 5, 4) == 1, "wait counts, NT = 4");
static_assert(lgkm_before(1, 0, 8) == 1 && lgkm_before(1, 3, 8) == 1 && lgkm_before(1, 5, 4) == 1, "wait counts, phase 1");
#endif

// NT = 16-column tiles per wave (8 | 4 | 1): the workgroup's tile is 256 x (32 NT).  Narrow outputs (the VAE's
// Cout = 128 and Cout = 3 layers, N <= 128 / N <= 32) take NT = 4 / 1: the two wave columns then share the real
// columns instead of one wave column doing everything and the other only padding, and the MFMAs of the missing
// tiles are not issued at all (compile-time: a run-time test per MFMA splits the loop into basic blocks and the
// compiler then drains the memory counters at every join - measured, 20 % slower than doing the padding).  The
// staging, the LDS reads and their slots are those of NT = 8.
// QKN kernels (the q | k | v and single-block in-projections): K2 - per-head RMSNorm of q and k, RoPE on the image rows,
// ref hunyuan_video_packed.py:137-160, 566-585 - applied in the epilogue, where a wave's 128 output columns are exactly one
// head and the row layout (16 lanes x 8 columns per row) is the layout norm_kernels.hip's K2 kernel works in: the same
// arithmetic on the same bf16-rounded values, so the result is bit-identical to the separate pass, which is then not run.
struct QkNormArgs {
    const bf16 *wq, *wk, *wq_txt, *wk_txt;      // RMSNorm weights [128]; *_txt (may be null): rows >= rope_rows
    const float* rope;                          // [rope_rows][cos 128 | sin 128]
    int rope_rows, qcol0, kcol0, width;         // q / k occupy output columns [qcol0, +width) / [kcol0, +width)
    float eps;
};
// TALL = 1 (the VAE's Cout = 128 convolutions at full resolution): the workgroup's tile is 512 x 128 - the four waves
// stacked, each with 128 rows x all 128 columns, i.e. the wave-level loop of NT = 8 (64 MFMAs per phase, 32 fragment
// reads per 128 MFMAs) where the 256 x 128 tile of NT = 4 gives a wave 128 x 64 (24 reads per 64 MFMAs) and stages
// 48 KB per 1 024 matrix cycles; this one stages 80 KB per 2 048.  LDS (all 160 KiB): wave w's A rows are private to
// it - [w][buffer][128 rows] at w * 32 KiB - and staged by itself (8 pieces per half unit), W [buffer][128 rows] above
// them (4 pieces per wave); the epilogue's row stage is the wave's own A-h1 half of buffer 1, which the next tile's
// prologue does not fill (A-h1(1) is issued in the next tile's first phase, behind its barrier).
// GRP = 1 (the double-stream blocks' linears, ref hunyuan_video_packed.py:537-604: the image rows and the text rows of one
// activation buffer go through two nn.Linear of the same shape): rows [0, m_split) use (W, bias, gate), rows
// [m_split, M) use Rows2's (w2, bias2, gate2); a row tile lies in one segment (the second segment's tiles start at
// m_split), and the text segment's 4 - 16 row tiles ride in the image segment's partly filled last round instead of
// taking a launch - a round of a 256-CU chip - of their own.
struct Rows2 {
    const void *w2, *bias2, *gate2;
    int m_split;
};
template <typename T, typename TO, int EPI, int AMODE, int NT = 8, int QKN = 0, int TALL = 0, int GRP = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const T* __restrict__ A, int64_t lda, int64_t a2_delta,
                                                         int k1, const T* __restrict__ W, int64_t ldw,
                                                         const TO* __restrict__ bias_, TO* __restrict__ C,
                                                         int64_t ldc, int M, int N, int K, int gelu_from,
                                                         const TO* __restrict__ R, int64_t ldr,
                                                         const TO* __restrict__ gate_, int ntm, int ntn, ConvGeom cg,
                                                         const float* __restrict__ a_scale,
                                                         const float* __restrict__ w_scale, QkNormArgs qk, Rows2 r2) {
    static_assert(!GRP || (AMODE == 0 && !TALL && !QKN && NT == 8 && sizeof(T) == 2), "row segments: the dense 16-bit 256 x 256 tile");
    typedef i32x4 frag_t;
    typedef TO Tx4 __attribute__((ext_vector_type(4)));
    constexpr bool SCALED = sizeof(T) == 1;
    constexpr int BKE = 128 / (int)sizeof(T);
    constexpr int LOG2_TB = sizeof(T) == 1 ? 0 : 1;  // log2 of an element's bytes
    constexpr int KS = mfma_op<T>::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // A[2] 32 KiB each | W[2] | stage 32 KiB
    static_assert(!TALL || (NT == 8 && KS == 2 && QKN == 0 && sizeof(TO) == 2), "the tall tile: 16-bit, all 8 column tiles per wave");
    constexpr unsigned W_BASE = TALL ? 4 * OP_BYTES : 2 * OP_BYTES;
    constexpr unsigned A_STEP = TALL ? OP_BYTES / 2 : OP_BYTES;      // from an operand's buffer 0 to its buffer 1
    constexpr unsigned W_STEP = TALL ? OP_BYTES / 2 : OP_BYTES;
    constexpr int TBM = TALL ? 2 * BM : BM;                          // tile rows
    constexpr int APW = TALL ? 8 : 4;                                // DMA pieces per wave: of an A half unit,
    constexpr int WPW = TALL ? 4 : NT;                               // of a W unit

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = TALL ? wave : wave >> 1, wc = TALL ? 0 : wave & 1;

    // ---- DMA sources.  One instruction = 8 tile rows x 128 B.  A half unit = 16 instructions (4 per wave),
    //      W unit = 32 instructions (8 per wave).
    const int prow = lane >> 3, pch = lane & 7;
    // per-lane byte offsets of the DMA / load rows from the tile's first row (the tile base and the K offset are
    // wave-uniform and stay in scalar registers: the loads take the address as scalar base + 32-bit lane offset)
    unsigned pa[2][APW], pw[8];
    const char *abase = nullptr, *wbase = nullptr;
    int arow0[2][APW], wrow0[8];
    unsigned adst[2][APW];                      // LDS offset of the piece in buffer 0 (wave-uniform)
    // AMODE 1 (convolution gather), per DMA row of the tile: coff = source of the CENTRE tap in BYTES from the tile's
    // scalar base `abase` (= the start of the earliest frame any tap of the tile's first row reads: every source of the
    // tile lies at or behind it, within 32 bits - the launcher checks; the lane's swizzled chunk included), so that a
    // piece's address is scalar base + 32-bit lane offset like the dense operand's: no 64-bit vector arithmetic per piece; ccode = nine 2-bit signed voxel steps, [2 i] for the time
    // taps dt = i - 2, [6 + 2 i] for dh = i - 1, [12 + 2 i] for dw = i - 1: how far the clamped (replicate padding,
    // causal in time), stride-folded source coordinate of that tap lies from the centre's, each in {-2 .. 1}.
    // A K tile's source is then three field extracts and two multiply-adds away (uniform tap), instead of the
    // clamps, shifts and 64-bit multiplies per piece and K tile that made the gather the longest stream of the loop.
    unsigned coff[2][APW], ccode[2][APW];
    int m0 = 0, n0 = 0;
    int m_end = M;                              // GRP: the end of the tile's row segment (else M)
    const T* w_seg = W;                         // GRP: the segment's weights
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            if (TALL) {                                                // the wave's own 128 rows
                arow0[h][j] = wave * 128 + h * 64 + j * 8;
                adst[h][j] = (unsigned)wave * (2 * A_STEP) + (unsigned)(h * 64 + j * 8) * 128;
            } else {
                const int i = wave * 4 + j;                            // 0..15
                arow0[h][j] = (i >> 3) * 128 + h * 64 + (i & 7) * 8;
                adst[h][j] = (unsigned)arow0[h][j] * 128;
            }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) wrow0[j] = (wave * WPW + j) * 8;     // WPW pieces per wave cover the tile's rows of W

    // tile order: XCD-contiguous, grouped 8 row-tiles tall so that co-running tiles share A and W panels.  The
    // workgroup is persistent: it takes tiles vb = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 or
    // the whole tile count, so vb & 7 stays this workgroup's XCD).
    const int ntiles = ntm * ntn;
    constexpr int CW = NT * 16;     // columns per wave column
    unsigned a_rd[2], w_rd[2];  // fragment read addresses (below)
    const int fr = lane & 15, fq = lane >> 4;
    constexpr unsigned SECOND = SCALED ? 16u : 64u;
    const unsigned off_k0 = (unsigned)(fr * 128 + (((SCALED ? 2 * fq : fq) ^ (fr >> 1)) << 4));
    auto setup = [&](int vb) __attribute__((always_inline)) {
        // ConvGeom::gm: gm > 0: gm row tiles x all column tiles per group; gm < 0: -gm column tiles x all row tiles
        const int GM = cg.gm != 0 ? abs(cg.gm) : 8;
        const int wid = xcd_remap(vb, ntiles);
        const int nmaj = cg.gm < 0 ? ntn : ntm, nmin = cg.gm < 0 ? ntm : ntn;
        const int per_group = GM * nmin;
        const int group = wid / per_group, in_group = wid % per_group;
        const int first = group * GM;
        const int gsz = min(nmaj - first, GM);
        const int tmaj = first + in_group % gsz, tmin = in_group / gsz;
        int tm = cg.gm < 0 ? tmin : tmaj;
        const int tn = cg.gm < 0 ? tmaj : tmin;
        if (AMODE == 1 && cg.band > 0) {               // band-major walk (ConvGeom::band): position -> (band, frame, tile)
            const int per = cg.band * cg.To, bnd = tm / per, rem = tm - bnd * per;
            const int t = rem / cg.band;
            tm = t * cg.tf + bnd * cg.band + (rem - t * cg.band);
        }
        m0 = tm * TBM;
        if constexpr (GRP) {
            const int ntm0 = (r2.m_split + TBM - 1) / TBM;
            const bool second = tm >= ntm0;
            m0 = second ? r2.m_split + (tm - ntm0) * TBM : m0;
            m_end = second ? M : r2.m_split;
            w_seg = second ? reinterpret_cast<const T*>(r2.w2) : W;
        }
        n0 = tn * (TALL ? CW : 2 * CW);
        int lo_vox = 0;                             // AMODE 1: first voxel of the frame the tile's earliest tap reads (uniform)
        if (AMODE == 1) {
            const int tc = (m0 / (cg.Wo * cg.Ho)) << cg.dt, x = max(tc - 2, 0);
            lo_vox = (cg.st ? ((x + 1) >> 1) : x) * (cg.Hi * cg.Wi);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int ar = arow0[h][j] + prow;
                if (AMODE == 0) {
                    pa[h][j] = (unsigned)(min(m0 + ar, m_end - 1) - m0) * (unsigned)(lda * sizeof(T)) + (pch ^ ((ar >> 1) & 7)) * 16;
                } else {
                    const int m = min(m0 + ar, M - 1);
                    const int wc = (m % cg.Wo) << cg.dw, hc = ((m / cg.Wo) % cg.Ho) << cg.dh,
                              tc = (m / (cg.Wo * cg.Ho)) << cg.dt;
                    auto src_t = [&](int x) { x = max(x, 0); return cg.st ? ((x + 1) >> 1) : x; };
                    auto src_h = [&](int y) { return min(max(y, 0), cg.hmax) >> cg.sh; };
                    auto src_w = [&](int x) { return min(max(x, 0), cg.wmax) >> cg.sw; };
                    const int t0 = src_t(tc), h0 = src_h(hc), w0 = src_w(wc);
                    unsigned code = 0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        code |= (unsigned)((src_t(tc + i - 2) - t0) & 3) << (2 * i);
                        code |= (unsigned)((src_h(hc + i - 1) - h0) & 3) << (6 + 2 * i);
                        code |= (unsigned)((src_w(wc + i - 1) - w0) & 3) << (12 + 2 * i);
                    }
                    ccode[h][j] = code;
                    // the row's swizzle term is (prow >> 1) | 4 (j & 1)  (rows come 8 per instruction)
                    const unsigned chunk = (unsigned)(pch ^ (prow >> 1) ^ ((j & 1) << 2));
                    coff[h][j] = ((unsigned)((t0 * cg.Hi + h0) * cg.Wi + w0 - lo_vox) << (cg.lcin + LOG2_TB)) + chunk * 16;
                }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int wrw = wrow0[j] + prow;
            pw[j] = (unsigned)(min(n0 + wrw, N - 1) - n0) * (unsigned)(ldw * sizeof(T)) + (pch ^ ((wrw >> 1) & 7)) * 16;
        }
        abase = AMODE == 1 ? reinterpret_cast<const char*>(A) + ((int64_t)lo_vox << (cg.lcin + LOG2_TB))
                           : reinterpret_cast<const char*>(A + (int64_t)m0 * lda);
        wbase = reinterpret_cast<const char*>(w_seg + (int64_t)n0 * ldw);

    };
    const int nk = K / BKE;
    // AMODE 1: a K tile is one tap x 64 or 128 channels, and with Cin >= 2 K tiles per tap (the launcher sends narrower
    // inputs to the 8-wave kernel) the tiles 2 i and 2 i + 1 share their tap: the piece's source offset is worked out
    // at the EVEN tile and kept for the odd one (poff), which halves the gather's vector instructions - a lone wave per
    // SIMD pays an issue slot for every one of them.  Which call is which is known at compile time (below).
    unsigned poff[2][APW];
    auto tap_off = [&](int h, int j, int t) __attribute__((always_inline)) -> unsigned {
        const unsigned k0 = (unsigned)t * BKE;
        const unsigned tap = k0 >> cg.lcin;                                      // (uniform; tap < 27)
        const unsigned t3 = (tap * 11u) >> 5, it = (tap * 57u) >> 9;             // tap / 3, tap / 9
        const unsigned ih = t3 - 3u * it, iw = tap - 3u * t3;
        const int code = (int)ccode[h][j];
        const int dT = __builtin_amdgcn_sbfe(code, 2 * it, 2), dH = __builtin_amdgcn_sbfe(code, 6 + 2 * ih, 2),
                  dW = __builtin_amdgcn_sbfe(code, 12 + 2 * iw, 2);
        const int dvox = __mul24(dT, cg.Hi * cg.Wi) + __mul24(dH, cg.Wi) + dW;
        return coff[h][j] + ((unsigned)dvox << (cg.lcin + LOG2_TB));
    };
    // The same offset in two halves of three vector instructions each, for the main loop: a lone wave per SIMD issues
    // one instruction per ~4 cycles, so whatever stands between two MFMAs beyond ~3 instructions idles the matrix pipe
    // (16 cycles per MFMA); worked out in one place, a piece's offset + its issue were 10 - 12 instructions.  The halves
    // go to the two MFMA gaps in front of the piece's own (`slot`).  tap_bits: the tap's bit positions in ccode (uniform,
    // once per phase).
    int tapT = 0, tapH = 0, tapW = 0;
    unsigned ptmp = 0;
    auto tap_bits = [&](int t) __attribute__((always_inline)) {
        t = min(t, nk - 1);
        const unsigned tap = ((unsigned)t * BKE) >> cg.lcin;
        const unsigned t3 = (tap * 11u) >> 5, it = (tap * 57u) >> 9;
        tapT = (int)(2 * it);
        tapH = (int)(6 + 2 * (t3 - 3u * it));
        tapW = (int)(12 + 2 * (tap - 3u * t3));
    };
    auto prep_a = [&](auto PART, int h, int j) __attribute__((always_inline)) {
        const int code = (int)ccode[h][j];
        if constexpr (PART == 0) {
            ptmp = (unsigned)(__mul24(__builtin_amdgcn_sbfe(code, tapH, 2), cg.Wi) + __builtin_amdgcn_sbfe(code, tapW, 2));
        } else {
            const int dvox = __mul24(__builtin_amdgcn_sbfe(code, tapT, 2), cg.Hi * cg.Wi) + (int)ptmp;
            poff[h][j] = coff[h][j] + ((unsigned)dvox << (cg.lcin + LOG2_TB));
        }
    };
    auto src_a = [&](int h, int j, int t, auto EVEN) __attribute__((always_inline)) -> const T* {
        if (AMODE == 0) {
            const int64_t koff = (int64_t)t * BKE + ((t * BKE >= k1) ? a2_delta : 0);
            asm volatile("" : "+v"(pa[h][j]));      // keeps "base + lane offset" from being hoisted as a 64-bit register pair
            return reinterpret_cast<const T*>(abase + koff * (int64_t)sizeof(T) + pa[h][j]);
        } else {
            const unsigned c0 = ((unsigned)t * BKE) & (unsigned)(cg.Cin - 1);        // (uniform)
            if constexpr (EVEN) poff[h][j] = tap_off(h, j, t);
            asm volatile("" : "+v"(poff[h][j]));    // (as above: scalar base + 32-bit lane offset)
            return reinterpret_cast<const T*>(abase + (int64_t)(c0 * sizeof(T)) + poff[h][j]);
        }
    };
    auto src_w = [&](int j, int t) __attribute__((always_inline)) -> const T* {
        asm volatile("" : "+v"(pw[j]));
        return reinterpret_cast<const T*>(wbase + (int64_t)t * (BKE * (int)sizeof(T)) + pw[j]);
    };
    // EVEN: t is even before the clamp (clamped, an even and the odd call behind it both name tile nk - 1)
#if defined(FP_PROBE_REG_LOAD)
    i32x4 probe_sink = {0, 0, 0, 0};
#endif
    // PAR (main loop of the gather, AMODE 1): the destination buffer as the compile-time parity of the UNCLAMPED tile
    // index - a load past the last K tile then lands in the buffer that tile would have used, which nobody reads any
    // more (its last reader finished behind the previous barrier; W fragments are in registers) - so the LDS address
    // is base + immediate instead of three scalar instructions on the clamped index.  -1: the clamped index's parity.
    auto issue_a1 = [&](int h, int j, int t, auto EVEN, auto PAR) __attribute__((always_inline)) {
#if defined(FP_PROBE_NO_A_DMA)      // timing-only probe builds (results wrong): what the A operand's LDS-DMA pieces cost
        if (AMODE == 1 && (FP_PROBE_NO_A_DMA == 1 || j != 0)) return;      // 1: none of them; 2: one of four (what shared W taps would leave: a third)
#endif
#if defined(FP_PROBE_NO_DMA)        // the same for the dense operands (mask: 1 = no A pieces, 2 = no W pieces)
        if (AMODE == 0 && (FP_PROBE_NO_DMA & 1)) return;
#endif
#if defined(FP_PROBE_REG_LOAD)      // timing only: the same loads into registers (never stored to LDS) instead of LDS-DMA
        if (AMODE == 0) {
            t = min(t, nk - 1);
            const int64_t koff = (int64_t)t * BKE + ((t * BKE >= k1) ? a2_delta : 0);
            const char* sb = abase + koff * (int64_t)sizeof(T);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(probe_sink) : "v"(pa[h][j]), "s"(sb));
            return;
        }
#endif
        t = min(t, nk - 1);
        const unsigned par = decltype(PAR)::value >= 0 ? (unsigned)decltype(PAR)::value : (unsigned)(t & 1);
        dma16(src_a(h, j, t, EVEN), smem + par * A_STEP + adst[h][j]);
    };
    auto issue_w1 = [&](int j, int t, auto PAR) __attribute__((always_inline)) {
#if defined(FP_PROBE_NO_DMA)
        if (AMODE == 0 && (FP_PROBE_NO_DMA & 2)) return;
#endif
#if defined(FP_PROBE_REG_LOAD)
        if (AMODE == 0) {
            t = min(t, nk - 1);
            const char* sb = wbase + (int64_t)t * (BKE * (int)sizeof(T));
            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(probe_sink) : "v"(pw[j]), "s"(sb));
            return;
        }
#endif
        t = min(t, nk - 1);
        const unsigned par = decltype(PAR)::value >= 0 ? (unsigned)decltype(PAR)::value : (unsigned)(t & 1);
        dma16(src_w(j, t), smem + W_BASE + par * W_STEP + wrow0[j] * 128);
    };

    // the order the counted waits assume: A-h0(0), W(0) | A-h1(0) | A-h0(1), W(1)   (3 APW + 2 WPW loads per wave)
    auto prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < APW; ++j) issue_a1(0, j, 0, ic<1>{}, ic<-1>{});
#pragma unroll
        for (int j = 0; j < WPW; ++j) issue_w1(j, 0, ic<-1>{});
#pragma unroll
        for (int j = 0; j < APW; ++j) issue_a1(1, j, 0, ic<1>{}, ic<-1>{});
#pragma unroll
        for (int j = 0; j < APW; ++j) issue_a1(0, j, 1, ic<0>{}, ic<-1>{});
#pragma unroll
        for (int j = 0; j < WPW; ++j) issue_w1(j, 1, ic<-1>{});
    };

    // ---- fragment read offsets (as gemm_pipeline.h)
    // One address register per K step and operand; every other offset - the buffer's included: the two A buffers
    // are neighbours, and so are the two W buffers - is an instruction immediate below 64 KiB.
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_rd[ks] = (TALL ? (unsigned)wave * (2 * A_STEP) : (unsigned)(wr * 128) * 128) + (ks ? (off_k0 ^ SECOND) : off_k0);
        w_rd[ks] = W_BASE + (unsigned)(wc * CW) * 128 + (ks ? (off_k0 ^ SECOND) : off_k0);
        asm volatile("" : "+v"(a_rd[ks]), "+v"(w_rd[ks]));
    }

    // The 64 accumulator tiles live in AGPRs a[0:255] that the compiler never sees: tile (mi, ni) = a[(mi*8+ni)*4 .. +3].
    // hipcc cannot keep 256 loop-carried accumulator registers in place once the MFMAs are interleaved with loads:
    // whether they come from the MFMA builtin or from "+a" asm operands, it renames tiles and copies them through
    // VGPRs (hundreds of v_accvgpr moves per K tile, or spills).  So the MFMAs and the final reads are inline asm
    // on literal register numbers, the reservation below makes the kernel descriptor count the AGPRs, and the
    // kernel keeps its VGPR need below 256 so that the compiler never reaches for an AGPR as spill space
    // (tests/test_host_cpu.py checks the built code object for that).  The first K step of a tile multiplies onto
    // a zero C operand: no zero fill.
    reserve_agprs();

    // W fragments of the current K tile: [ks][ni].  A fragments: a ring of 4 groups; group g of a phase =
    // (ks = g >> 2, mi = 4 p + (g & 3)).
    frag_t wf[KS == 2 ? 2 : 1][8], af[A_RING];
    // fp8 (KS == 1): a tile's operand is the pair of 16-B reads side by side.  A fragments of BOTH phases' row halves
    // (the held-back MFMAs of a phase still read the previous half's), W fragments of the K tile.
    i32x8 a8[KS == 1 ? 2 : 1][4], w8[KS == 1 ? 8 : 1];
    auto rd_a8 = [&](auto BUF, auto P, auto MI) __attribute__((always_inline)) {
        const unsigned char* p = smem + BUF * OP_BYTES + (P * 64 + MI * 16) * 128;
        a8[KS == 1 ? (int)P : 0][MI] = __builtin_shufflevector(*reinterpret_cast<const frag_t*>(p + a_rd[0]),
                                                               *reinterpret_cast<const frag_t*>(p + a_rd[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto rd_w8 = [&](auto BUF, auto NI) __attribute__((always_inline)) {
        const unsigned char* p = smem + BUF * OP_BYTES + (NI * 16) * 128;
        w8[KS == 1 ? (int)NI : 0] = __builtin_shufflevector(*reinterpret_cast<const frag_t*>(p + w_rd[0]),
                                                            *reinterpret_cast<const frag_t*>(p + w_rd[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // MFMA (row tile MI of half P, column tile NI); ZC: the tile's only K step of this K tile is its first
    auto mfma8 = [&](auto P, auto MI, auto NI, auto ZC) __attribute__((always_inline)) {
        mfma8_acc<((P * 4 + MI) * 8 + NI) * 4, ZC>(w8[KS == 1 ? (int)NI : 0], a8[KS == 1 ? (int)P : 0][MI]);
    };

    // The 16-bit loop's fragment reads are inline asm, and every wait for them is written out (lgkm_before below).
    // As C++ loads they were waited for by the compiler, which does not see the loop's own `s_waitcnt` in front of
    // the barrier: it still held the previous phase's last two reads for pending, and drained the counter - with the
    // new phase's first ten reads just issued - in front of the held-back MFMAs that were meant to cover their latency.
    // (The dynamic LDS segment starts at address 0: the kernel has no static LDS.)
    auto rd_a = [&](auto BUF, auto P, auto G) __attribute__((always_inline)) {
        constexpr int off = BUF * (int)A_STEP + (P * 64 + (G & 3) * 16) * 128;
        lds_read_b128<off>(af[G & (A_RING - 1)], a_rd[G >> 2]);
    };
    auto rd_w = [&](auto BUF, auto KS_, auto NI) __attribute__((always_inline)) {
        constexpr int off = BUF * (int)W_STEP + (NI * 16) * 128;
        if constexpr (NI < NT) lds_read_b128<off>(wf[KS_][NI], w_rd[KS_]);
    };
    // MFMA ni of group G of phase P; ZC: first K step of the tile (C operand = 0); LGKM: wait in front (-1: none)
    auto mfma1 = [&](auto P, auto G, auto NI, auto ZC, auto LGKM) __attribute__((always_inline)) {
        constexpr int MI = P * 4 + (G & 3);
        if constexpr (NI < NT) mfma_acc<T, (MI * 8 + NI) * 4, (ZC && (G >> 2) == 0), LGKM>(wf[G >> 2][NI], af[G & (A_RING - 1)]);
    };
    // what is issued behind MFMA k of group g: at most one LDS read or one LDS-DMA per MFMA gap.
    //   P0(t) issues A-h1(t+1);  P1(t) issues A-h0(t+2) and W(t+2)  (W's fragments sit in registers after P0).
    auto slot = [&](auto BUF, auto P, int t, auto G, auto K_) __attribute__((always_inline)) {
        constexpr int g = G, k = K_;
        constexpr int g4 = g < 4 ? g : 0;
        if constexpr (k == 0 && g + A_AHEAD < 8) rd_a(BUF, P, ic<g + A_AHEAD>{});
        // AMODE 1: the gathered pieces' offsets are due when the tile they load is an even one (t + 1 in buffer 1's P0,
        // t + 2 in buffer 0's P1); their two halves go to the gaps in front of the piece's
        constexpr bool PREP = AMODE == 1 && ((P == 0 && BUF == 1) || (P == 1 && BUF == 0));
        constexpr int PAR1 = AMODE == 1 ? 1 - BUF : -1, PAR2 = AMODE == 1 ? (int)BUF : -1;      // buffers of tiles t + 1, t + 2
        if constexpr (P == 0) {
            if constexpr (g < 4 && k == 1) rd_w(BUF, ic<1>{}, ic<2 * g4>{});
            if constexpr (g < 4 && k == 2) rd_w(BUF, ic<1>{}, ic<2 * g4 + 1>{});
            if constexpr (PREP && g < 4 && k == 1) prep_a(ic<0>{}, 1, g4);
            if constexpr (PREP && g < 4 && k == 2) prep_a(ic<1>{}, 1, g4);
            if constexpr (PREP && TALL && g < 4 && k == 3) prep_a(ic<0>{}, 1, 4 + g4);
            if constexpr (PREP && TALL && g < 4 && k == 5) prep_a(ic<1>{}, 1, 4 + g4);
            if constexpr (g < 4 && k == 4) issue_a1(1, g, t + 1, ic<0>{}, ic<PAR1>{});
            if constexpr (TALL && g < 4 && k == 6) issue_a1(1, 4 + g4, t + 1, ic<0>{}, ic<PAR1>{});
        } else {
            if constexpr (PREP && g < 4 && k == 0) prep_a(ic<0>{}, 0, g4);
            if constexpr (PREP && g < 4 && k == 1) prep_a(ic<1>{}, 0, g4);
            if constexpr (PREP && TALL && g < 4 && k == 3) prep_a(ic<0>{}, 0, 4 + g4);
            if constexpr (PREP && TALL && g < 4 && k == 4) prep_a(ic<1>{}, 0, 4 + g4);
            if constexpr (k == 2) {
                if constexpr (g < 4) issue_a1(0, g, t + 2, ic<0>{}, ic<PAR2>{});
                else if constexpr (g - 4 < WPW) issue_w1(g - 4, t + 2, ic<PAR2>{});
            }
            if constexpr (k == 5 && 2 + g < WPW) issue_w1(2 + g, t + 2, ic<PAR2>{});
            if constexpr (TALL && g < 4 && k == 6) issue_a1(0, 4 + g4, t + 2, ic<0>{}, ic<PAR2>{});
        }
    };
    auto fence = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    // relaxed: the 32 output stores of the previous tile may still be in flight behind the loads this phase needs
    // (memory operations retire in order), so the count that guarantees the loads is 16 + 32.
    // PZC (fp8 only): ZC of the PREVIOUS phase, whose held-back tiles this phase issues (in the 16-bit loop the
    // held-back groups are second K steps and never start a tile)
    auto phase = [&](auto BUF, auto P, int t, auto HAVE_PREV, auto ZC, bool relaxed, auto PZC) __attribute__((always_inline)) {
        // younger than the unit a phase needs are always 2 APW + WPW pieces (APW of A-h0, APW of A-h1, WPW of W)
        static_assert(2 * APW + WPW + 32 <= 63, "vmcnt is a 6-bit counter");
        if (relaxed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * APW + WPW + 32) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * APW + WPW) : "memory");
        __builtin_amdgcn_s_barrier();
        fence();
        if constexpr (KS == 1) {
            // fp8: 32 MFMAs of 32 cycles per phase, column tile n outermost (its W fragment arrives two tiles
            // ahead, in the gaps), the half's four A fragments read up front; tiles n = 6, 7 are held back behind
            // the next barrier like the 16-bit loop's last two groups.
            if constexpr (P == 0) {
                rd_w8(BUF, ic<0>{});
                rd_w8(BUF, ic<1>{});
            }
            static_for<4>([&](auto mi) __attribute__((always_inline)) { rd_a8(BUF, P, mi); });
            fence();
            if constexpr (HAVE_PREV) {
                static_for<2>([&](auto n) __attribute__((always_inline)) {
                    static_for<4>([&](auto mi) __attribute__((always_inline)) { mfma8(ic<1 - P>{}, mi, ic<6 + n>{}, PZC); });
                });
            }
            fence();
            static_for<6>([&](auto n) __attribute__((always_inline)) {
                static_for<4>([&](auto mi) __attribute__((always_inline)) {
                    mfma8(P, mi, n, ZC);
                    fence();
                    if constexpr (P == 0) {
                        if constexpr (mi == 0) rd_w8(BUF, ic<n + 2>{});
                        if constexpr (mi == 2 && n < 4) issue_a1(1, n, t + 1, ic<(BUF == 1)>{}, ic<-1>{});
                    } else {
                        if constexpr (mi == 0) {
                            if constexpr (n < 4) issue_a1(0, n, t + 2, ic<(BUF == 0)>{}, ic<-1>{});
                            else issue_w1(n - 4, t + 2, ic<-1>{});
                        }
                        if constexpr (mi == 2) issue_w1(2 + n, t + 2, ic<-1>{});
                    }
                    fence();
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if constexpr (P == 0) static_for<8>([&](auto ni) __attribute__((always_inline)) { rd_w(BUF, ic<0>{}, ni); });
            static_for<A_AHEAD>([&](auto g) __attribute__((always_inline)) { rd_a(BUF, P, g); });
            if constexpr (AMODE == 1 && P == 0 && BUF == 1) tap_bits(t + 1);
            if constexpr (AMODE == 1 && P == 1 && BUF == 0) tap_bits(t + 2);
            fence();
            if constexpr (HAVE_PREV) {      // the two groups the previous phase left behind (fragments long landed)
                static_for<8>([&](auto ni) __attribute__((always_inline)) { mfma1(ic<1 - P>{}, ic<6>{}, ni, ic<0>{}, ic<-1>{}); });
                static_for<8>([&](auto ni) __attribute__((always_inline)) { mfma1(ic<1 - P>{}, ic<7>{}, ni, ic<0>{}, ic<-1>{}); });
            }
            fence();
            static_for<6>([&](auto g) __attribute__((always_inline)) {
                static_for<8>([&](auto k) __attribute__((always_inline)) {
                    mfma1(P, g, k, ZC, ic<(k == 0 ? lgkm_before(P, g, NT) : -1)>{});
                    fence();
                    slot(BUF, P, t, g, k);
                    fence();
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // ---- epilogue of the tile at (em0, en0): lane (fr,fq) holds row em0 + wr*128 + mi*16 + fr,
    //      columns en0 + wc*128 + ni*16 + 4 fq + (0..3)
    // Output rows leave through a wave-private LDS stage (the 32 KiB above the operand buffers, 2 x 4 KiB per
    // wave): the accumulator layout (lane = row, 4 columns) is rewritten as rows of 256 B, so every global store -
    // and every residual load - is 16 B per lane and covers whole 128-byte lines (4 rows x 256 B per instruction).
    // Stage image: [16 rows][256 B], 16-B chunk index XORed with the row (conflict-free b128 reads, 2-way b64 writes).
    typedef TO Tx8 __attribute__((ext_vector_type(8)));
    Tx8 rpre[2][4];          // EPI 3/4: residual rows (row layout) of two row tiles, a ring filled 2 tiles ahead
    // (the epilogue derives its lane terms from an opaque copy of the lane number: otherwise the compiler computes
    //  some fifty per-lane offsets at kernel entry and holds them in registers across every main loop)
    auto opaque_lane = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    // (GRP: the epilogue runs after the NEXT tile's setup, so the segment's terms are derived again from the tile's row)
    auto seg_rows = [&](int em0) __attribute__((always_inline)) { return GRP ? (em0 >= r2.m_split ? M : r2.m_split) : M; };
    auto load_resid = [&](int em0, int en0, int ecw, int mi) __attribute__((always_inline)) {
        const int M = seg_rows(em0);            // shadows the kernel's
        const int el = opaque_lane();
        const int gcol = (el & 15) * 8 < ecw ? en0 + wc * ecw + (el & 15) * 8 : N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = em0 + wr * 128 + mi * 16 + (el >> 4) + 4 * i;
            rpre[mi & 1][i] = Tx8{};
            if (gcol < N && m < M) rpre[mi & 1][i] = *reinterpret_cast<const Tx8*>(R + (int64_t)m * ldr + gcol);
        }
    };
    auto epilogue = [&](int em0, int en0, int ecw) __attribute__((always_inline)) {
      const int M = seg_rows(em0);              // shadows the kernel's
      const bool second = GRP && em0 >= r2.m_split;
      const TO* const bias = second ? reinterpret_cast<const TO*>(r2.bias2) : bias_;
      const TO* const gate = second ? reinterpret_cast<const TO*>(r2.gate2) : gate_;
      if constexpr (sizeof(TO) == 4) {
        // fp32 output (the VAE mid-block attention's logits, the DiT's fp32 head; EPI 0, optional fp32 bias): straight
        // from the accumulators, 16 B per lane - a row tile's store covers 16 rows x 64 B per instruction.  N % 4 == 0.
        const int el = opaque_lane();
        const int fr = el & 15, fq = el >> 4;
        const int nb = en0 + wc * ecw + 4 * fq;
        f32x4 bv[8];
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias && ni < NT) bv[ni] = *reinterpret_cast<const f32x4*>(bias + min(nb + ni * 16, N - 4));
        }
        static_for<8>([&](auto MI_) __attribute__((always_inline)) {
            constexpr int mi = MI_;
            const int mrow0 = em0 + wr * 128 + mi * 16;
            if (mrow0 >= M) return;                            // wave-uniform
            static_for<8>([&](auto NI_) __attribute__((always_inline)) {
                constexpr int ni = NI_;
                if constexpr (ni >= NT) return;
                const int n = nb + ni * 16;
                const f32x4 av = acc_read<(mi * 8 + ni) * 4>() + bv[ni];
                if (mrow0 + fr < M && n < N) *reinterpret_cast<f32x4*>(C + (int64_t)(mrow0 + fr) * ldc + n) = av;
            });
        });
      } else {
        // (TALL: the wave's own A-h1 half of buffer 1 - see the kernel's comment)
        unsigned stage_off = TALL ? (unsigned)wave * (2 * A_STEP) + A_STEP + 8192u : 2 * BUF_BYTES + wave * 8192;      // opaque, so that the 4096 of the second stage buffer
        asm volatile("" : "+v"(stage_off));                    // stays an instruction immediate, not 12 more addresses
        unsigned char* stage = smem + stage_off;
        Tx4 bv[8], gv[8];
        f32x4 ws[8];
        const int el = opaque_lane();
        const int fr = el & 15, fq = el >> 4;                  // shadow the main loop's
        const int nb = en0 + wc * ecw + 4 * fq;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const int n = min(nb + ni * 16, N - 4);
            bv[ni] = Tx4{0, 0, 0, 0};
            if (bias) bv[ni] = *reinterpret_cast<const Tx4*>(bias + n);
            if (EPI == 3) gv[ni] = *reinterpret_cast<const Tx4*>(gate + n);
            if (SCALED) ws[ni] = *reinterpret_cast<const f32x4*>(w_scale + n);
        }
        const int lrow = el >> 4, lchk = el & 15;
        const int gcol = en0 + wc * ecw + lchk * 8;            // first of this lane's 8 columns in row layout
        const bool col_ok = gcol < N && lchk * 8 < ecw;        // N % 8 == 0: a chunk is inside or outside
        // QKN: is this wave's column block a q head (0), a k head (1) or neither (-1)?  (uniform)
        int qk_which = -1;
        bf16x8 qk_w_img = {}, qk_w_txt = {};
        if constexpr (QKN) {
            const int c0 = __builtin_amdgcn_readfirstlane(en0 + wc * ecw);
            if (c0 >= qk.qcol0 && c0 < qk.qcol0 + qk.width) qk_which = 0;
            else if (c0 >= qk.kcol0 && c0 < qk.kcol0 + qk.width) qk_which = 1;
            if (qk_which >= 0) {
                const bf16* wi = qk_which ? qk.wk : qk.wq;
                const bf16* wt = qk_which ? qk.wk_txt : qk.wq_txt;
                qk_w_img = *reinterpret_cast<const bf16x8*>(wi + lchk * 8);
                qk_w_txt = wt ? *reinterpret_cast<const bf16x8*>(wt + lchk * 8) : qk_w_img;
            }
        }
        static_for<8>([&](auto MI_) __attribute__((always_inline)) {
            constexpr int mi = MI_;
            unsigned char* sb = stage + (mi & 1) * 4096;
            const int mrow0 = em0 + wr * 128 + mi * 16;
            if (mrow0 >= M) return;                            // wave-uniform
            if (EPI == 3 || EPI == 4) {                        // residual rows: registers (row layout) -> stage
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lrow + 4 * i;
                    *reinterpret_cast<Tx8*>(sb + row * 256 + ((lchk ^ row) << 4)) = rpre[mi & 1][i];
                }
                if (mi + 2 < 8) load_resid(em0, en0, ecw, mi + 2);
            }
            const float as = SCALED ? a_scale[min(mrow0 + fr, M - 1)] : 1.f;
            // QKN: the four rows' cos | sin go out before the accumulators are converted (their latency is covered by it)
            f32x4 qk_cs[QKN ? 4 : 1][4];
            if constexpr (QKN) {
                if (qk_which >= 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = mrow0 + lrow + 4 * i;
                        if (m < qk.rope_rows) {
                            const float* cs = qk.rope + (int64_t)m * 256 + lchk * 8;
                            qk_cs[i][0] = *reinterpret_cast<const f32x4*>(cs);
                            qk_cs[i][1] = *reinterpret_cast<const f32x4*>(cs + 4);
                            qk_cs[i][2] = *reinterpret_cast<const f32x4*>(cs + 128);
                            qk_cs[i][3] = *reinterpret_cast<const f32x4*>(cs + 132);
                        }
                    }
                }
            }
            static_for<8>([&](auto NI_) __attribute__((always_inline)) {
                constexpr int ni = NI_;
                if constexpr (ni >= NT) return;
                const int n = nb + ni * 16;
                unsigned char* sp = sb + fr * 256 + ((((ni * 2) + (fq >> 1)) ^ fr) << 4) + (fq & 1) * 8;
                const f32x4 av = acc_read<(mi * 8 + ni) * 4>();
                const bool do_gelu = EPI == 1 || (EPI == 2 && n >= gelu_from);
                float v[4];
                if (SCALED) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = av[j] * (as * ws[ni][j]) + (float)bv[ni][j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = av[j] + (float)bv[ni][j];
                }
                if (do_gelu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_tanh(v[j]);
                }
                if (EPI == 3 || EPI == 4) {
                    const Tx4 rv = *reinterpret_cast<const Tx4*>(sp);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)rv[j] + (EPI == 3 ? (float)gv[ni][j] : 1.f) * v[j];
                }
                Tx4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (TO)v[j];
                *reinterpret_cast<Tx4*>(sp) = o;
            });
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lrow + 4 * i;
                Tx8 ov = *reinterpret_cast<const Tx8*>(sb + row * 256 + ((lchk ^ row) << 4));
                if constexpr (QKN) {
                    // K2's arithmetic (norm_kernels.hip, qk_rmsnorm_rope_kernel), rounding point by rounding point
                    if (qk_which >= 0) {
                        const bool is_txt = mrow0 + row >= qk.rope_rows;
                        const bf16x8 wv = is_txt ? qk_w_txt : qk_w_img;
                        float f[8], ss = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            f[j] = (float)ov[j];
                            ss += f[j] * f[j];
                        }
                        const float r = rsqrtf(group16_sum(ss) * (1.f / 128.f) + qk.eps);
                        float nn[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const bf16 a = (bf16)(f[j] * r);
                            nn[j] = (float)(bf16)((float)a * (float)wv[j]);
                        }
                        if (!is_txt) {
                            const f32x4 c0 = qk_cs[i][0], c1 = qk_cs[i][1], s0 = qk_cs[i][2], s1 = qk_cs[i][3];
                            const float c[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                            const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
                            for (int p2 = 0; p2 < 4; ++p2) {
                                ov[2 * p2] = (TO)(nn[2 * p2] * c[2 * p2] + (-nn[2 * p2 + 1]) * sn[2 * p2]);
                                ov[2 * p2 + 1] = (TO)(nn[2 * p2 + 1] * c[2 * p2 + 1] + nn[2 * p2] * sn[2 * p2 + 1]);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) ov[j] = (TO)nn[j];
                        }
                    }
                }
                if (col_ok && mrow0 + row < M) *reinterpret_cast<Tx8*>(C + (int64_t)(mrow0 + row) * ldc + gcol) = ov;
            }
        });
      }
    };

    static_assert(KS == 2 || (KS == 1 && NT == 8), "16-bit MFMA 16x16x32 (two K steps per tile) or fp8 16x16x128 (one)");
    int vb = blockIdx.x;
    setup(vb);
    prologue();
    bool relaxed = false;
    for (;;) {
        const int em0 = m0, en0 = n0;
        constexpr int ecw = CW;
        phase(ic<0>{}, ic<0>{}, 0, ic<0>{}, ic<1>{}, relaxed, ic<0>{});
        phase(ic<0>{}, ic<1>{}, 0, ic<1>{}, ic<1>{}, relaxed, ic<1>{});
        int t = 1;
        if (nk > 1) {
            phase(ic<1>{}, ic<0>{}, 1, ic<1>{}, ic<0>{}, relaxed, ic<1>{});
            phase(ic<1>{}, ic<1>{}, 1, ic<1>{}, ic<0>{}, false, ic<0>{});
            for (t = 2; t + 1 < nk; t += 2) {
                phase(ic<0>{}, ic<0>{}, t, ic<1>{}, ic<0>{}, false, ic<0>{});
                phase(ic<0>{}, ic<1>{}, t, ic<1>{}, ic<0>{}, false, ic<0>{});
                phase(ic<1>{}, ic<0>{}, t + 1, ic<1>{}, ic<0>{}, false, ic<0>{});
                phase(ic<1>{}, ic<1>{}, t + 1, ic<1>{}, ic<0>{}, false, ic<0>{});
            }
            if (t < nk) {
                phase(ic<0>{}, ic<0>{}, t, ic<1>{}, ic<0>{}, false, ic<0>{});
                phase(ic<0>{}, ic<1>{}, t, ic<1>{}, ic<0>{}, false, ic<0>{});
            }
        }
        fence();
        if constexpr (KS == 1) {
            auto tail = [&](auto PZC) __attribute__((always_inline)) {
                static_for<2>([&](auto n) __attribute__((always_inline)) {
                    static_for<4>([&](auto mi) __attribute__((always_inline)) { mfma8(ic<1>{}, mi, ic<6 + n>{}, PZC); });
                });
            };
            if (nk == 1) tail(ic<1>{});          // the only K tile's held-back tiles start from C = 0
            else tail(ic<0>{});
        } else {
            static_for<8>([&](auto ni) __attribute__((always_inline)) { mfma1(ic<1>{}, ic<6>{}, ni, ic<0>{}, ic<-1>{}); });
            static_for<8>([&](auto ni) __attribute__((always_inline)) { mfma1(ic<1>{}, ic<7>{}, ni, ic<0>{}, ic<-1>{}); });
        }
        fence();
        // every wave is done with the LDS (and the clamped tail loads have landed): the next tile's first loads
        // go out BEFORE this tile's epilogue, whose stores then drain under the next main loop
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (EPI == 3 || EPI == 4) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) load_resid(em0, en0, ecw, mi);
        }
        fence();
        vb += gridDim.x;
        const bool more = vb < ntiles;
        if (more) {
            setup(vb);
            prologue();
        }
        fence();
        asm volatile("s_nop 15\ns_nop 15" ::: "memory");        // the last MFMAs retire before their tiles are read
        epilogue(em0, en0, ecw);
#if defined(FP_PROBE_REG_LOAD)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(probe_sink));
#endif
        if (!more) break;
        // the operand pointers are recomputed rather than kept across the epilogue (32 VGPRs it needs itself)
        asm volatile("" : "+s"(vb));
        setup(vb);
        if (AMODE == 1) {                      // (not carried across the epilogue either: phase 0 reuses tile 0's for A-h1)
#pragma unroll
            for (int j = 0; j < APW; ++j) poff[1][j] = tap_off(1, j, 0);
        }
        // a ragged tile skips stores, so the relaxed count would not cover the loads: drain instead
        if (em0 + TBM > seg_rows(em0) || en0 + (TALL ? CW : 2 * CW) > N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        relaxed = true;
    }
}

// one workgroup per CU (the LDS footprint allows no more); a multiple of 8 so that a workgroup's tiles stay on its XCD
inline int persistent_grid() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        n = cus / 8 * 8;
    }
    return n;
}

template <typename T, int EPI, int AMODE, typename TO = T, int NT = 8, int QKN = 0, int TALL = 0, int GRP = 0>
int launch_w4(const void* a, int64_t lda, int64_t a2_delta, int k1, const void* w, int64_t ldw, const void* bias,
              void* c, int64_t ldc, int M, int N, int K, int gelu_from, const void* resid, int64_t ldr,
              const void* gate, const ConvGeom& cg, hipStream_t st, const float* a_scale = nullptr,
              const float* w_scale = nullptr, const QkNormArgs& qk = QkNormArgs{}, const Rows2& r2 = Rows2{}) {
    // lane offsets inside a tile are 32-bit: 255 rows x the row stride in bytes has to fit
    if (lda >= (int64_t(1) << 22) || ldw >= (int64_t(1) << 22)) return (int)hipErrorInvalidValue;
    constexpr int TBM = TALL ? 2 * BM : BM, TBN = TALL ? 16 * NT : 32 * NT;
    const int ntm = GRP ? (r2.m_split + TBM - 1) / TBM + (M - r2.m_split + TBM - 1) / TBM : (M + TBM - 1) / TBM;
    const int ntn = (N + TBN - 1) / TBN;
    // operand buffers + the epilogue's row stage (TALL: all 160 KiB are operand buffers, the stage lies inside them)
    constexpr int smem = TALL ? 5 * OP_BYTES : 2 * BUF_BYTES + 4 * 8192;
    auto kern = gemm_w4_kernel<T, TO, EPI, AMODE, NT, QKN, TALL, GRP>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    kern<<<dim3((unsigned)min(ntm * ntn, persistent_grid())), dim3(256), smem, st>>>((const T*)a, lda, a2_delta, k1, (const T*)w, ldw,
                                                              (const TO*)bias, (TO*)c, ldc, M, N, K, gelu_from,
                                                              (const TO*)resid, ldr, (const TO*)gate, ntm, ntn, cg,
                                                              a_scale, w_scale, qk, r2);
    FP_CHECK_LAUNCH();
    return 0;
}

}  // namespace fp_gemm
