// K1: varlen flash-attention forward, bf16, head_dim 128, non-causal  (gfx950 MFMA 32x32x16, wave64)
//
// Replaces ref diffusers_helper/models/hunyuan_video_packed.py:89-112 (attn_varlen_func: the SDPA
// branch :97 is the parity target; the sageattention / flash / xformers dispatch is gone).
//
// Structure (one workgroup = 8 waves = 256 query rows of one head; each wave owns 32 query rows):
//   * Q fragments stay in registers for the whole KV sweep (32 VGPRs).
//   * K and V tiles of 64 keys are staged global -> registers -> LDS, double buffered.
//   * S^T = K.Q^T (operands swapped): the 32x32 accumulator then has the QUERY on the lane and 16 keys in
//     registers, so the online softmax is lane-local (one cross-half exchange per tile for the max).
//   * O^T = V^T.P^T: the S^T accumulators, converted to bf16 in place, ARE the B operand (no LDS round
//     trip); V^T fragments come from LDS through ds_read_b64_tr_b16 (hardware transpose), with the k
//     index permuted to match the accumulator's register->key map.
//   * K image: [key][256 B] with the 16-B chunk index XOR (key & 15)  -> conflict-free ds_read_b128.
//     V image: [d/32][key][64 B]                                      -> conflict-free tr reads.
//   * Tails: rows are clamped on load, scores of keys past the segment end are masked to -inf in the
//     last tile only, query rows past the end are not stored.  Nothing is padded in memory.
#include <mutex>
#include <type_traits>
#include "fp_common.h"

namespace {

constexpr int KVBLK = 64;     // keys per tile
constexpr int HD = 128;       // head dim
constexpr int KBUF = KVBLK * HD * 2;   // 16 KiB per K (or V) buffer
constexpr int WS_FLOATS = 256 * HD + 2 * 256;   // one partial sweep: O[256][128], m[256], l[256] (fp32)

__device__ __forceinline__ unsigned k_lds_off(int key, int chunk) {      // chunk = 16-B piece of the 256-B row
    return (unsigned)(key * 256 + ((chunk ^ (key & 15)) << 4));
}
__device__ __forceinline__ unsigned v_lds_off(int key, int chunk) {
    return (unsigned)((chunk >> 2) * (KVBLK * 64) + key * 64 + ((chunk & 3) << 4));
}

// Schedule.  Each wave's tile step is split into a vector segment V_t (online softmax of S(t) -> P(t)) and a matrix
// segment M_t (O += V(t)^T.P(t), then S(t+1) = K(t+1).Q^T), with a workgroup barrier after each.  Waves 4-7
// run one barrier behind waves 0-3, and waves w and w+4 share a SIMD: while one wave of a SIMD is in its matrix
// segment the other is in its vector segment, so the matrix pipe and the vector ALU overlap instead of both
// waves doing MFMAs and then both doing exp2 (the lockstep the one-barrier-per-tile loop falls into).
// K runs two tiles ahead of V in LDS (M_t needs K(t+1) and V(t)); the global loads for K(t+2) / V(t+1) are
// issued at the start of V_t and written to LDS at the end of M_t.
// value of x in lane (lane ^ 32), without LDS: v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// Diagnostic build only (-DFP_ATTN_DIAG, see tools/attn_diag.py): s_memtime stamps around the segments of the
// tile loop, summed per wave into fp_attn_dbg by workgroup 8.  No stamp executes in the shipped build.
#ifdef FP_ATTN_DIAG
__device__ unsigned long long fp_attn_dbg[8 * 8];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FP_STAMP(var) var = stamp()
#define FP_DIAG(...) __VA_ARGS__
#else
#define FP_STAMP(var)
#define FP_DIAG(...)
#endif

template <int THR_LOG2>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                             const bf16* __restrict__ V, bf16* __restrict__ O,
                                                             int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                                                             const int* __restrict__ cu_seqlens, int nseg, int total,
                                                             int heads, int nqb, float scale_log2e, int nwg_full,
                                                             int split, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* kbuf = smem;                 // [2][KBUF]
    unsigned char* vbuf = smem + 2 * KBUF;      // [2][KBUF]

    constexpr int QBLK = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 31, hh = lane >> 5;

    // Work items [0, nwg_full) are whole (head, q block) sweeps, dealt so that consecutive items of one XCD share
    // a head (K/V through its L2).  The items of the last, partial round are each cut into `split` KV ranges
    // (blocks >= nwg_full) that write partial (O, m, l) to `ws` for attn_combine_kernel: the chip stays full.
    int w, part = 0, nparts = 1;
    if ((int)blockIdx.x < nwg_full) {
        w = xcd_remap(blockIdx.x, nwg_full);
    } else {
        const int si = blockIdx.x - nwg_full;
        w = nwg_full + si / split;
        part = si % split;
        nparts = split;
    }
    const int qb = w % nqb;
    const int head = (w / nqb) % heads;
    const int seg = w / (nqb * heads);
    int seg_lo = 0, seg_hi = total;
    if (cu_seqlens) {
        seg_lo = cu_seqlens[seg];
        seg_hi = cu_seqlens[seg + 1];
    }
    const int q0 = seg_lo + qb * QBLK;
    if (q0 >= seg_hi) return;                   // uniform for the workgroup
    const int nkv = seg_hi - seg_lo;
    const int nt_all = (nkv + KVBLK - 1) / KVBLK;
    const int t0 = (int)((int64_t)nt_all * part / nparts);          // this block's KV tiles: [t0, t0 + nt)
    const int nt = (int)((int64_t)nt_all * (part + 1) / nparts) - t0;
    float* wsp = ws + (int64_t)((w - nwg_full) * split + part) * WS_FLOATS;
    if (nt == 0) {                              // (uniform) more parts than tiles: a neutral partial
        for (int i = tid; i < 256; i += 512) {
            wsp[256 * HD + i] = -INFINITY;
            wsp[256 * HD + 256 + i] = 0.f;
        }
        return;
    }

    const int qrow = min(q0 + wave * 32 + r, seg_hi - 1);
    bf16x8 qf[8];
    {
        const bf16* qp = Q + (int64_t)qrow * q_rs + head * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // staging: keys sk0 and sk0 + 32 of a tile, 16-B chunk sc.  Addresses are uniform base + 32-bit byte offset
    // (the launcher checks that an operand spans < 4 GiB); the offsets advance by one tile per step and are
    // clamped to the segment's last row, so no 64-bit multiply sits in the loop.
    const int sk0 = tid >> 4, sc = tid & 15;
    const unsigned k_step = (unsigned)(KVBLK * k_rs * 2), v_step = (unsigned)(KVBLK * v_rs * 2);
    const unsigned col_b = (unsigned)((head * HD + sc * 8) * 2);
    const unsigned k_max = (unsigned)((seg_hi - 1) * k_rs * 2) + col_b, v_max = (unsigned)((seg_hi - 1) * v_rs * 2) + col_b;
    unsigned k_cur[2], v_cur[2];                // byte offsets of this thread's two rows in the NEXT tile to load
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        k_cur[i] = (unsigned)((seg_lo + t0 * KVBLK + sk0 + 32 * i) * k_rs * 2) + col_b;
        v_cur[i] = (unsigned)((seg_lo + t0 * KVBLK + sk0 + 32 * i) * v_rs * 2) + col_b;
    }
    u32x4 kreg[2], vreg[2];
    auto load_k = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            kreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(K) + min(k_cur[i], k_max));
            k_cur[i] += k_step;
        }
    };
    auto load_v = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            vreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(V) + min(v_cur[i], v_max));
            v_cur[i] += v_step;
        }
    };
    auto write_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(kbuf + buf * KBUF + k_lds_off(sk0 + 32 * i, sc)) = kreg[i];
    };
    auto write_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(vbuf + buf * KBUF + v_lds_off(sk0 + 32 * i, sc)) = vreg[i];
    };

    const int i16 = lane & 15;
    const unsigned v_lane_off = (unsigned)((4 * hh + (i16 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2);

    f32x16 o_acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) o_acc[d][j] = 0.f;
    f32x16 s_acc[2];
    bf16x8 pf[4];
    float m_run = -INFINITY, l_run = 0.f;

    auto qk = [&](int buf) {                    // S^T(tile in kbuf[buf]) = K.Q^T
        const unsigned char* kp = kbuf + buf * KBUF;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 16; ++j) s_acc[kb][j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kp + k_lds_off(kb * 32 + r, 2 * ks + hh));
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s_acc[kb], 0, 0, 0);
            }
        }
    };

    // ---- prologue: K(0), V(0), K(1) into LDS; S(0)
    load_k();
    load_v();
    write_k(0);
    write_v(0);
    load_k();                                   // K(1) (a clamped repeat when nt == 1)
    write_k(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    qk(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();

    FP_DIAG(unsigned long long T0, Ta, Tb, Tc, T1, T2, T3, T4, T5, aLd = 0, aMx = 0, aRs = 0, aV = 0, aB1 = 0, aM = 0, aW = 0, aB2 = 0;)
    for (int t = 0; t < nt; ++t) {
        // ================= vector segment V_t
        FP_STAMP(T0);
        // unconditional (tile index clamped): a load under a runtime condition makes the compiler merge old
        // and new register values right behind the load, i.e. wait for it at once
        load_k();                               // K(t+2)
        load_v();                               // V(t+1)

        if (t0 + t == nt_all - 1 && (nkv & (KVBLK - 1))) {
            const int valid = nkv - (t0 + t) * KVBLK;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int key = kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                    if (key >= valid) s_acc[kb][j] = -INFINITY;
                }
        }
        FP_STAMP(Ta);
        float mx = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s_acc[kb][j]);
        mx = fmaxf(mx, other_half(mx)) * scale_log2e;
        FP_DIAG(asm volatile("" ::"v"(mx));)
        FP_STAMP(Tb);
        if (!__all(mx - m_run <= (float)THR_LOG2)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int j = 0; j < 16; ++j) o_acc[d][j] *= alpha;
        }
        FP_DIAG(asm volatile("" ::"v"(o_acc[0]), "v"(o_acc[3]));)
        FP_STAMP(Tc);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][j], scale_log2e, -m_run));
                rs += p;
                pf[kb * 2 + (j >> 3)][j & 7] = (bf16)p;
            }
        l_run += rs;

        FP_DIAG(asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(l_run));)
        FP_STAMP(T1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        FP_STAMP(T2);

        // ================= matrix segment M_t
        // Eight groups of four MFMAs: O^T d-blocks 0..3 (V^T fragments: 8 transposed reads each), then
        // S^T(t+1) as (key block, k-step quad) x 4 (K fragments: 4 ds_read_b128 each).  The LDS reads of
        // group g+2 are issued before the MFMAs of group g (three fragment sets), so that only the first
        // group's LDS latency is exposed; the scheduling barriers keep hipcc from undoing that order.
        __builtin_amdgcn_s_setprio(1);          // the matrix segment's issue slots come first
        {
            const unsigned char* vp = vbuf + (t & 1) * KBUF + v_lane_off;
            const unsigned char* kp = kbuf + ((t + 1) & 1) * KBUF;
            // S^T(t+1) is computed after the last tile too (from a stale K buffer, result unused): keeping the
            // segment branch-free lets the LDS-read counts of consecutive groups chain exactly.
            bf16x8 fa[3][4];
            auto rd_v = [&](bf16x8(&f)[4], int d) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    union {
                        s16x4 h[2];
                        bf16x8 v;
                    } a;
                    a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vp + d * (KVBLK * 64) + (16 * s) * 64));
                    a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vp + d * (KVBLK * 64) + (16 * s + 8) * 64));
                    f[s] = a.v;
                }
            };
            auto rd_k = [&](bf16x8(&f)[4], int kb, int kq) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    f[i] = *reinterpret_cast<const bf16x8*>(kp + k_lds_off(kb * 32 + r, 2 * (4 * kq + i) + hh));
            };
            // group g: 0..3 = O^T d-block g ; 4..7 = S^T key block (g-4)>>1, k-step quad (g-4)&1
            auto rd = [&](bf16x8(&f)[4], int g) {
                if (g < 4) rd_v(f, g);
                else rd_k(f, (g - 4) >> 1, (g - 4) & 1);
            };
            rd(fa[0], 0);
            rd(fa[1], 1);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 2 < 8) rd(fa[(g + 2) % 3], g + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (g < 4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        o_acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g % 3][s], pf[s], o_acc[g], 0, 0, 0);
                } else {
                    const int kb = (g - 4) >> 1, kq = (g - 4) & 1;
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i)       // the chain's first MFMA takes the constant 0 as C (no register clear)
                        s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g % 3][i], qf[4 * kq + i],
                                                                            (kq == 0 && i == 0) ? zero : s_acc[kb], 0, 0, 0);
                }
                if (g == 2) {
                    // the tiles loaded at the start of V_t go to the LDS buffers nobody reads in this segment
                    if (t + 2 < nt) write_k(t & 1);
                    if (t + 1 < nt) write_v((t + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        FP_DIAG(asm volatile("" ::"v"(s_acc[0]), "v"(s_acc[1]), "v"(o_acc[0]), "v"(o_acc[3]));)
        FP_STAMP(T3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FP_STAMP(T4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        FP_STAMP(T5);
        FP_DIAG(aV += T1 - Tc; aLd += Ta - T0; aMx += Tb - Ta; aRs += Tc - Tb; aB1 += T2 - T1; aM += T3 - T2;
                aW += T4 - T3; aB2 += T5 - T4;)
    }
    FP_DIAG(if (blockIdx.x == 8 && lane == 0) {
        unsigned long long* d = fp_attn_dbg + wave * 8;
        d[0] = aV; d[1] = aB1; d[2] = aM; d[3] = aW; d[4] = aB2; d[5] = nt; d[6] = aLd; d[7] = (aMx << 32) | aRs;
    })
    if (grp == 0) __builtin_amdgcn_s_barrier();

    const float l_tot = l_run + other_half(l_run);
    if (nparts > 1) {                           // partial sweep: unnormalised O (fp32), running max and sum
        const int row = wave * 32 + r;
        float* op = wsp + row * HD + 4 * hh;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o_acc[d][4 * g + j];
                *reinterpret_cast<f32x4*>(op + d * 32 + g * 8) = v;
            }
        if (hh == 0) {
            wsp[256 * HD + row] = m_run;
            wsp[256 * HD + 256 + row] = l_tot;
        }
        return;
    }
    const float inv = 1.f / l_tot;
    const int qrow_real = q0 + wave * 32 + r;
    if (qrow_real < seg_hi) {
        bf16* op = O + (int64_t)qrow_real * o_rs + head * HD + 4 * hh;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (bf16)(o_acc[d][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(op + d * 32 + g * 8) = v;
            }
    }
}

// ======================================================================================================
// Second-generation sweep: ONE wave per SIMD (4 waves = the same 256 query rows per workgroup), each wave owning
// 64 query rows (two 32-row blocks rb).  Every K / V^T fragment read from LDS then feeds two MFMAs, and there is no
// partner wave to hand the matrix pipe to: the wave interleaves its own softmax VALU, LDS reads and LDS-DMA issues
// one small item per MFMA gap (an MFMA 32x32x16 holds the pipe for 32 cycles, 24 of which take filler issue).
//
//   iteration t:   MFMA stream   slots  0..31  O^T += V(t-1)^T . P(t-1)^T      (P of the PREVIOUS tile: no wait on the softmax)
//                                slots 32..63  S^T(t+1) = K(t+1) . Q^T         -> the AGPR half S(t) is not in
//                  VALU stream   slots  0..7   S(t): AGPR -> VGPR
//                                slots  8..13  row max, defer-max test (rescale factors if the max grew by > 2^THR)
//                                slots 14..63  P(t) = exp2(S(t) * scale - m), row sums, bf16 pack: 32 items of two values,
//                                              each behind the P.V MFMAs that used the fragment it overwrites
//                  after slot 63               O *= alpha if the max grew (rare; after V(t-1) is in)
//   Measured (tools/attn_clock_probe.py and builds with one stream removed): the VECTOR stream is the longer one -
//   a tile's 64 exp2 (16 cycles each), 64 fma, 64 add, 64 AGPR reads, 36 max, 32 packs and the address / LDS / DMA
//   issue take ~3 200 cycles against 2 048 cycles of MFMA; with the MFMAs removed a sweep takes 82 % of the full
//   time, with the softmax removed 67 %.
//   Accumulators: O^T (4 d-blocks x 2 rb) = a[0:127], S^T (2 key blocks x 2 rb, two tiles) = a[128:255], addressed by number
//   from inline asm (the compiler cannot keep them in place; see gemm_w4.h).  S(t) (64 VGPRs), P (32), Q (64).
//   K and V tiles come by LDS-DMA into two rings of three 16 KiB slots (K three tiles ahead, V one); the per-lane
//   SOURCE address carries the K image's XOR swizzle.  One barrier and one counted vmcnt(8) per tile.
template <int N> using ic = std::integral_constant<int, N>;
template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic<I>{});
        static_for<N, I + 1>(f);
    }
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void dma16(const void* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
}
// a[IDX:IDX+15] += A . B after at most LGKM LDS reads are left in flight (the V^T fragments are read by inline asm -
// see rd_v - so the compiler does not count them)
template <int IDX, int LGKM> __device__ __forceinline__ void mfma32_after(const bf16x8& a, const bf16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(IDX), "n"(IDX + 15),
                 "v"(a), "v"(b), "n"(LGKM));
}
// a[IDX:IDX+15] (+)= A . B
template <int IDX, bool ZERO_C> __device__ __forceinline__ void mfma32(const bf16x8& a, const bf16x8& b) {
    if constexpr (ZERO_C)
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, 0" ::"n"(IDX), "n"(IDX + 15), "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(IDX), "n"(IDX + 15), "v"(a), "v"(b));
}
template <int IDX> __device__ __forceinline__ float agpr_read() {
    float r;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(r) : "n"(IDX));
    return r;
}
template <int IDX> __device__ __forceinline__ void agpr_write(float v) {
    asm volatile("v_accvgpr_write_b32 a[%0], %1" ::"n"(IDX), "v"(v));
}
// (lo, hi) -> packed bf16 pair.  NOT inline asm: gfx950 needs a wait state between a transcendental (the exp2 that
// produces lo/hi) and a VALU instruction that consumes its result, which the compiler only inserts in front of
// instructions it knows; as asm, the pack read stale values in some lanes.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2 r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

// Softmax item i (0..31) of a tile runs behind MFMA 14 + (50 i) / 32; the P^T fragment it completes (k-step i / 8) has
// been consumed by the previous tile's P.V MFMAs (slots 8 (i/8) .. 8 (i/8) + 7) by then.
constexpr int w64_item_of_slot(int m) {
    for (int i = 0; i < 32; ++i)
        if (14 + (50 * i) / 32 == m) return i;
    return -1;
}
// the k-th (0..7) slot from 14 on without a softmax item carries LDS-DMA instruction k
constexpr int w64_dma_of_slot(int m) {
    if (m < 14 || w64_item_of_slot(m) >= 0) return -1;
    int k = 0;
    for (int j = 14; j < m; ++j)
        if (w64_item_of_slot(j) < 0) ++k;
    return k < 8 ? k : -1;
}

template <int THR_LOG2>
__global__ __launch_bounds__(256, 1) void attn_w64_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, bf16* __restrict__ O,
                                                          int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                                                          const int* __restrict__ cu_seqlens, int nseg, int total,
                                                          int heads, int nqb, float scale_log2e, int nwg_full,
                                                          int split, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // K ring [3][16 KiB] | V ring [3][16 KiB]
    constexpr int VRING = 3 * KBUF;
    constexpr int QBLK = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    asm volatile("" ::: "a0", "a255");          // the kernel descriptor must count the AGPRs the asm below names

    int w, part = 0, nparts = 1;
    if ((int)blockIdx.x < nwg_full) {
        w = xcd_remap(blockIdx.x, nwg_full);
    } else {
        const int si = blockIdx.x - nwg_full;
        w = nwg_full + si / split;
        part = si % split;
        nparts = split;
    }
    const int qb = w % nqb;
    const int head = (w / nqb) % heads;
    const int seg = w / (nqb * heads);
    int seg_lo = 0, seg_hi = total;
    if (cu_seqlens) {
        seg_lo = cu_seqlens[seg];
        seg_hi = cu_seqlens[seg + 1];
    }
    const int q0 = seg_lo + qb * QBLK;
    if (q0 >= seg_hi) return;                   // uniform for the workgroup
    const int nkv = seg_hi - seg_lo;
    const int nt_all = (nkv + KVBLK - 1) / KVBLK;
    const int t0 = (int)((int64_t)nt_all * part / nparts);
    const int nt = (int)((int64_t)nt_all * (part + 1) / nparts) - t0;
    float* wsp = ws + (int64_t)((w - nwg_full) * split + part) * WS_FLOATS;
    if (nt == 0) {
        for (int i = tid; i < 256; i += 256) {
            wsp[256 * HD + i] = -INFINITY;
            wsp[256 * HD + 256 + i] = 0.f;
        }
        return;
    }

    bf16x8 qf[2][8];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int qrow = min(q0 + wave * 64 + rb * 32 + r, seg_hi - 1);
        const bf16* qp = Q + (int64_t)qrow * q_rs + head * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[rb][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- LDS-DMA staging.  K: instruction i of this wave = keys wave*16 + 4 i + (0..3), lane = (key, 16-B position);
    //      the lane fetches the chunk that belongs at its position under the image's XOR.  V: instruction i = d-block
    //      `wave`, keys 16 i + (0..15), lane = (key, 16-B piece of the 64-B row).
    const unsigned kcolb[2] = {(unsigned)((head * HD + ((lane & 15) ^ ((lane >> 4) & 15)) * 8) * 2),
                               (unsigned)((head * HD + ((lane & 15) ^ ((4 + (lane >> 4)) & 15)) * 8) * 2)};
    // key & 15 of instruction i, lane l: (4 i + (l >> 4)) & 15 -> i = 0..3 gives 0-3, 4-7, 8-11, 12-15
    const unsigned kxor_i[4] = {0u, 4u, 8u, 12u};
    (void)kcolb;
    const unsigned k_step = (unsigned)(KVBLK * k_rs * 2), v_step = (unsigned)(KVBLK * v_rs * 2);
    const unsigned k_rowmax = (unsigned)((seg_hi - 1) * k_rs * 2), v_rowmax = (unsigned)((seg_hi - 1) * v_rs * 2);
    const unsigned k_i = (unsigned)(4 * k_rs * 2), v_i = (unsigned)(16 * v_rs * 2);
    unsigned k_row = (unsigned)((seg_lo + t0 * KVBLK + wave * 16 + (lane >> 4)) * k_rs * 2);     // next K tile to load
    unsigned v_row = (unsigned)((seg_lo + t0 * KVBLK + (lane >> 2)) * v_rs * 2);                 // next V tile to load
    const unsigned v_col = (unsigned)((head * HD + wave * 32 + (lane & 3) * 8) * 2);
    const unsigned k_pos = lane & 15, k_key = lane >> 4;
    auto issue_k = [&](int i, int slot) __attribute__((always_inline)) {
        const unsigned col = (unsigned)((head * HD + ((k_pos ^ ((kxor_i[i] + k_key) & 15)) * 8)) * 2);
        const unsigned off = min(k_row + i * k_i, k_rowmax) + col;
        dma16(reinterpret_cast<const unsigned char*>(K) + off, smem + slot * KBUF + (wave * 16 + i * 4) * 256);
    };
    auto issue_v = [&](int i, int slot) __attribute__((always_inline)) {
        const unsigned off = min(v_row + i * v_i, v_rowmax) + v_col;
        dma16(reinterpret_cast<const unsigned char*>(V) + off, smem + VRING + slot * KBUF + wave * (KVBLK * 64) + i * 1024);
    };

    // ---- fragment read addresses
    const int i16 = lane & 15;
    const unsigned v_lane_off = (unsigned)(VRING + (4 * hh + (i16 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2);
    const unsigned k_lane_row = (unsigned)(r * 256);
    const unsigned k_lane_x = (unsigned)(hh ^ (r & 15));       // chunk (2 ks + hh) ^ (r & 15) = (2 ks) ^ k_lane_x

    // ---- state
    float sv[2][2][16];                          // S(t): [rb][kb][j], key = kb*32 + (j&3) + 8*(j>>2) + 4*hh
    bf16x8 pf[2][4];                             // P(t-1)/P(t): [rb][k-step]
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[rb][s4][e] = (bf16)0.f;
    asm volatile(".set fp_i, 0\n.rept 128\nv_accvgpr_write_b32 a[fp_i], 0\n.set fp_i, fp_i + 1\n.endr" ::: "memory");

    auto fence = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    auto rd_k = [&](bf16x8& f, unsigned kbase, auto KB, auto KS_) __attribute__((always_inline)) {
        f = *reinterpret_cast<const bf16x8*>(smem + kbase + KB * 8192 + (((2 * KS_) ^ k_lane_x) << 4));
    };
    // Transposing reads as inline asm: behind an LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of the
    // ds_read_tr intrinsic (it cannot tell the ring slots apart), which would serialise the loads with the sweep.  The
    // asm reads are invisible to its lgkmcnt bookkeeping, so their consumers wait explicitly (mfma32_after).
    auto rd_v = [&](bf16x8& f, unsigned vbase, auto D, auto S4) __attribute__((always_inline)) {
        union {
            u32x2 h[2];
            bf16x8 v;
        } a;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a.h[0]) : "v"(vbase), "n"(D * (KVBLK * 64) + (16 * S4) * 64));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a.h[1]) : "v"(vbase), "n"(D * (KVBLK * 64) + (16 * S4 + 8) * 64));
        f = a.v;
    };

    // ---- prologue: K(0), K(1), V(0), K(2), V(0) again (so that "all but the last 8 loads" covers what a tile needs)
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(i, 0);
    k_row += k_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(i, 1);
    k_row += k_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_v(i, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(i, 2);
    k_row += k_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_v(i, 0);
    v_row += v_step;
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // K(0) landed
    __builtin_amdgcn_s_barrier();
    {   // S^T(0) = K(0) . Q^T -> a[128:191]
        const unsigned kbase = k_lane_row;
        static_for<8>([&](auto ks) __attribute__((always_inline)) {
            static_for<2>([&](auto kb) __attribute__((always_inline)) {
                bf16x8 f;
                rd_k(f, kbase, kb, ks);
                static_for<2>([&](auto rb) __attribute__((always_inline)) {
                    mfma32<128 + (kb * 2 + rb) * 16, (ks == 0)>(f, qf[rb][ks]);
                });
            });
        });
    }

    int kslot_rd = 1, kslot_wr = 0, vslot_rd = 0, vslot_wr = 1;      // for t = 0
    // One tile step.  PAR = t & 1: S(t) sits in a[128 + 64 PAR ..], S(t+1) is accumulated into the other half.
    auto step = [&](int t, auto PAR) __attribute__((always_inline)) {
        constexpr int SCUR = 128 + 64 * PAR, SNXT = 128 + 64 * (1 - PAR);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_nop 15\ns_nop 7" ::: "memory");               // the last MFMAs of S(t) have retired
        fence();
        const unsigned kbase = (unsigned)(kslot_rd * KBUF) + k_lane_row;
        const unsigned vbase = (unsigned)(vslot_rd * KBUF) + v_lane_off;
        const bool ragged_last = (t0 + t == nt_all - 1) && (nkv & (KVBLK - 1));
        const int valid = nkv - (t0 + t) * KVBLK;
        bf16x8 vf[4], kf[4];                                          // fragment rings (three fragments = 6 MFMAs ahead)
        float mx[2], rs[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f};
        bool grow = false;
        rd_v(vf[0], vbase, ic<0>{}, ic<0>{});
        rd_v(vf[1], vbase, ic<1>{}, ic<0>{});
        rd_v(vf[2], vbase, ic<2>{}, ic<0>{});
        fence();
        static_for<64>([&](auto M_) __attribute__((always_inline)) {
            constexpr int m = M_;
            // ---------------- the MFMA of this slot
            if constexpr (m < 32) {
                constexpr int s4 = m >> 3, d = (m >> 1) & 3, rb = m & 1, f = m >> 1;
                // still in flight behind fragment f: fragments f+1 and f+2 (two reads per V^T fragment, one per K fragment)
                if constexpr (rb == 0) mfma32_after<(d * 2 + rb) * 16, (f < 14 ? 4 : (f == 14 ? 3 : 2))>(vf[f & 3], pf[rb][s4]);
                else mfma32<(d * 2 + rb) * 16, false>(vf[f & 3], pf[rb][s4]);
            } else {
                constexpr int q = m - 32, ks = q >> 2, kb = (q >> 1) & 1, rb = q & 1, f = q >> 1;
                mfma32<SNXT + (kb * 2 + rb) * 16, (ks == 0)>(kf[f & 3], qf[rb][ks]);
            }
            fence();
            // ---------------- LDS reads: fragment f+3 while fragment f is multiplied (one fragment per two slots)
            if constexpr (m < 32 && (m & 1) == 0) {
                constexpr int f2 = (m >> 1) + 3;
                if constexpr (f2 < 16) rd_v(vf[f2 & 3], vbase, ic<f2 & 3>{}, ic<(f2 >> 2)>{});
                else rd_k(kf[f2 & 3], kbase, ic<(f2 - 16) & 1>{}, ic<((f2 - 16) >> 1)>{});
            }
            if constexpr (m >= 32 && (m & 1) == 0) {
                constexpr int f2 = ((m - 32) >> 1) + 3;
                if constexpr (f2 < 16) rd_k(kf[f2 & 3], kbase, ic<(f2 & 1)>{}, ic<(f2 >> 1)>{});
            }
            // ---------------- LDS-DMA: K(t+3) and V(t+1), one instruction in each of the first 8 slots from 14 on that
            // carry no softmax item
            if constexpr (w64_dma_of_slot(m) >= 0) {
                constexpr int i = w64_dma_of_slot(m);
                if constexpr (i < 4) issue_k(i, kslot_wr);
                else issue_v(i - 4, vslot_wr);
            }
            // ---------------- softmax items.  The vector ALU, not the matrix pipe, is the longer stream of a tile
            // (64 exp2 at 16 cycles each, plus ~300 plain instructions, plus 8 issue cycles per MFMA), so the items
            // are spread over ALL 64 MFMA gaps.
            if constexpr (m < 8) {               // S(t): AGPR -> VGPR, 8 values per slot
                constexpr int rb = m >> 2, kb = (m >> 1) & 1, j0 = (m & 1) * 8;
                static_for<8>([&](auto jj) __attribute__((always_inline)) {
                    sv[rb][kb][j0 + jj] = agpr_read<SCUR + (kb * 2 + rb) * 16 + j0 + jj>();
                });
                if constexpr (m == 7) {
                    if (ragged_last) {
#pragma unroll
                        for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
                            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                                for (int j = 0; j < 16; ++j) {
                                    const int key = kb2 * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                                    if (key >= valid) sv[rb2][kb2][j] = -INFINITY;
                                }
                    }
                }
            }
            if constexpr (m >= 8 && m < 12) {    // row max: 16 values per slot
                constexpr int i = m - 8, rb = i >> 1, kb = i & 1;
                float a0 = kb == 0 ? sv[rb][kb][0] : mx[rb];
                static_for<8>([&](auto jj) __attribute__((always_inline)) {
                    a0 = fmaxf(fmaxf(a0, sv[rb][kb][2 * jj]), sv[rb][kb][2 * jj + 1]);
                });
                mx[rb] = a0;
                asm volatile("" : "+v"(mx[rb]));
            }
            if constexpr (m == 12) {
                mx[0] = fmaxf(mx[0], other_half(mx[0])) * scale_log2e;
                mx[1] = fmaxf(mx[1], other_half(mx[1])) * scale_log2e;
                asm volatile("" : "+v"(mx[0]), "+v"(mx[1]));
            }
            if constexpr (m == 13) {
                grow = !__all((mx[0] - m_run[0] <= (float)THR_LOG2) && (mx[1] - m_run[1] <= (float)THR_LOG2));
                if (grow) {
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const float m_new = fmaxf(m_run[rb], mx[rb]);
                        alpha[rb] = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
                        m_run[rb] = m_new;
                        l_run[rb] *= alpha[rb];
                    }
                }
            }
            if constexpr (w64_item_of_slot(m) >= 0) {      // P(t): two values per item, in the order the P^T k-steps use them
                constexpr int i = w64_item_of_slot(m), s4 = i >> 3, rb = (i >> 2) & 1, dw = i & 3;
                constexpr int kb = s4 >> 1, j = (s4 & 1) * 8 + 2 * dw;
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[rb][kb][j], scale_log2e, -m_run[rb]));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[rb][kb][j + 1], scale_log2e, -m_run[rb]));
                rs[rb] += e0;
                rs[rb] += e1;
                union {
                    unsigned u[4];
                    bf16x8 v;
                } pk;
                pk.v = pf[rb][s4];
                unsigned pw = cvt_pk_bf16(e0, e1);
                // pin the item to its slot: volatile asm statements keep their order, and the MFMAs are such
                // statements (left alone, the compiler sinks the whole P computation behind the last MFMA)
                asm volatile("" : "+v"(pw), "+v"(rs[rb]));
                pk.u[dw] = pw;
                pf[rb][s4] = pk.v;
            }
            fence();
        });
        l_run[0] += rs[0];
        l_run[1] += rs[1];
        if (grow) {                                                   // O *= alpha, after V(t-1) has been added
            asm volatile("s_nop 15\ns_nop 15\ns_nop 7" ::: "memory");
            static_for<128>([&](auto I_) __attribute__((always_inline)) {
                constexpr int i = I_, rb = (i >> 4) & 1;
                agpr_write<i>(agpr_read<i>() * alpha[rb]);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        k_row += k_step;
        v_row += v_step;
        kslot_wr = kslot_rd;                                          // K(t+4) goes where K(t+1) was
        kslot_rd = kslot_rd == 2 ? 0 : kslot_rd + 1;
        vslot_rd = t == 0 ? 0 : (vslot_rd == 2 ? 0 : vslot_rd + 1);
        vslot_wr = vslot_wr == 2 ? 0 : vslot_wr + 1;
    };
    {
        int t = 0;
        for (; t + 1 < nt; t += 2) {
            step(t, ic<0>{});
            step(t + 1, ic<1>{});
        }
        if (t < nt) step(t, ic<0>{});
    }
    // ---- the last tile's P.V
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const unsigned vbase = (unsigned)(vslot_rd * KBUF) + v_lane_off;
        static_for<4>([&](auto s4) __attribute__((always_inline)) {
            static_for<4>([&](auto d) __attribute__((always_inline)) {
                bf16x8 f;
                rd_v(f, vbase, d, s4);
                static_for<2>([&](auto rb) __attribute__((always_inline)) { mfma32_after<(d * 2 + rb) * 16, 0>(f, pf[rb][s4]); });
            });
        });
    }
    asm volatile("s_nop 15\ns_nop 15\ns_nop 7" ::: "memory");

    float l_tot[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) l_tot[rb] = l_run[rb] + other_half(l_run[rb]);
    if (nparts > 1) {                           // partial sweep: unnormalised O (fp32), running max and sum
        static_for<2>([&](auto rb) __attribute__((always_inline)) {
            const int row = wave * 64 + rb * 32 + r;
            float* op = wsp + row * HD + 4 * hh;
            static_for<4>([&](auto d) __attribute__((always_inline)) {
                static_for<4>([&](auto g) __attribute__((always_inline)) {
                    f32x4 v;
                    static_for<4>([&](auto j) __attribute__((always_inline)) { v[(int)j] = agpr_read<(d * 2 + rb) * 16 + 4 * g + j>(); });
                    *reinterpret_cast<f32x4*>(op + d * 32 + g * 8) = v;
                });
            });
            if (hh == 0) {
                wsp[256 * HD + row] = m_run[rb];
                wsp[256 * HD + 256 + row] = l_tot[rb];
            }
        });
        return;
    }
    static_for<2>([&](auto rb) __attribute__((always_inline)) {
        const float inv = 1.f / l_tot[rb];
        const int qrow_real = q0 + wave * 64 + rb * 32 + r;
        if (qrow_real < seg_hi) {
            bf16* op = O + (int64_t)qrow_real * o_rs + head * HD + 4 * hh;
            static_for<4>([&](auto d) __attribute__((always_inline)) {
                static_for<4>([&](auto g) __attribute__((always_inline)) {
                    bf16x4 v;
                    static_for<4>([&](auto j) __attribute__((always_inline)) {
                        v[(int)j] = (bf16)(agpr_read<(d * 2 + rb) * 16 + 4 * g + j>() * inv);
                    });
                    *reinterpret_cast<bf16x4*>(op + d * 32 + g * 8) = v;
                });
            });
        }
    });
}

// Merge the `split` partial sweeps of each tail item: O = sum_i O_i 2^(m_i - M) / sum_i l_i 2^(m_i - M).
// One workgroup = 16 query rows x 16 chunks of 8 head-dim columns.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ ws, bf16* __restrict__ O,
                                                           int64_t o_rs, const int* __restrict__ cu_seqlens, int total,
                                                           int heads, int nqb, int nwg_full, int split) {
    const int item = nwg_full + blockIdx.x / 16;
    const int row = (blockIdx.x % 16) * 16 + (threadIdx.x >> 4), c8 = (threadIdx.x & 15) * 8;
    const int qb = item % nqb, head = (item / nqb) % heads, seg = item / (nqb * heads);
    int seg_lo = 0, seg_hi = total;
    if (cu_seqlens) {
        seg_lo = cu_seqlens[seg];
        seg_hi = cu_seqlens[seg + 1];
    }
    const int q = seg_lo + qb * 256 + row;
    if (q >= seg_hi) return;
    const float* base = ws + (int64_t)(blockIdx.x / 16) * split * WS_FLOATS;
    float M = -INFINITY;
    for (int i = 0; i < split; ++i) M = fmaxf(M, base[i * WS_FLOATS + 256 * HD + row]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, L = 0.f;
    for (int i = 0; i < split; ++i) {
        const float* p = base + i * WS_FLOATS;
        const float m = p[256 * HD + row];
        if (m == -INFINITY) continue;
        const float f = __builtin_amdgcn_exp2f(m - M);
        L += p[256 * HD + 256 + row] * f;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + row * HD + c8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + row * HD + c8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] += a[j] * f;
            acc[4 + j] += b[j] * f;
        }
    }
    const float inv = 1.f / L;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] * inv);
    bf16* op = O + (int64_t)q * o_rs + head * HD + c8;
    *reinterpret_cast<bf16x4*>(op) = bf16x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<bf16x4*>(op + 4) = bf16x4{o[4], o[5], o[6], o[7]};
}

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

}  // namespace

static int g_attn_variant = 1;      // 0: 8-wave ping-pong sweep | 1: one wave per SIMD, 64 query rows per wave
extern "C" int fp_debug_attn_variant(int v) {
    const int old = g_attn_variant;
    if (v >= 0) g_attn_variant = v;
    return old;
}

// Library-internal (attn_fp8.hip): the merge of split tail items and the CU count behind the split decision.
__attribute__((visibility("hidden"))) int fpi_attn_cu_count() { return cu_count(); }
__attribute__((visibility("hidden"))) int fpi_attn_combine(const float* ws, void* o, int64_t o_rs, int total, int heads,
                                                           int nqb, int nwg_full, int split, int ntail, hipStream_t st) {
    attn_combine_kernel<<<dim3((unsigned)(ntail * 16)), dim3(256), 0, st>>>(ws, (bf16*)o, o_rs, nullptr, total, heads, nqb,
                                                                            nwg_full, split);
    FP_CHECK_LAUNCH();
    return 0;
}

// Bytes of device workspace with which fp_attn_fwd_bf16 can split the last round's work items (any shape).
extern "C" int64_t fp_attn_workspace_bytes(void) { return (int64_t)cu_count() * WS_FLOATS * 4; }

// q, k, v, o: bf16 [total_tokens, heads, 128] with arbitrary row strides (elements) and head stride 128.
// cu_seqlens: device int32 [nseg+1] or NULL for a single segment [0, total_tokens).
// o may alias q (each workgroup reads its own Q rows before it writes them).
// workspace_required: a workspace too small for the split last round is an error instead of "run unsplit".
static int attn_launch(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride, int64_t k_row_stride,
                       int64_t v_row_stride, int64_t o_row_stride, const int* cu_seqlens, int nseg, int max_seqlen,
                       int total_tokens, int heads, int head_dim, float softmax_scale, void* workspace,
                       int64_t workspace_bytes, bool workspace_required, void* stream) {
    FP_REQUIRE(q && k && v && o && head_dim == HD && heads > 0 && total_tokens >= 0 && nseg >= 1);
    FP_REQUIRE(max_seqlen >= 0 && max_seqlen <= total_tokens);
    FP_REQUIRE(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && o_row_stride % 4 == 0);
    FP_REQUIRE(q_row_stride >= (int64_t)heads * HD && k_row_stride >= (int64_t)heads * HD &&
               v_row_stride >= (int64_t)heads * HD && o_row_stride >= (int64_t)heads * HD);
    FP_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 &&
               ((uintptr_t)o & 7) == 0);
    FP_REQUIRE(cu_seqlens || nseg == 1);
    // K and V are addressed as base + 32-bit byte offset (two tiles of run-ahead included)
    FP_REQUIRE((uint64_t)(total_tokens + 3 * KVBLK) * (uint64_t)k_row_stride * 2 < (1ull << 32) &&
               (uint64_t)(total_tokens + 3 * KVBLK) * (uint64_t)v_row_stride * 2 < (1ull << 32));
    if (total_tokens == 0 || max_seqlen == 0) return 0;
    const float scale_log2e = softmax_scale * 1.4426950408889634f;
    const int nqb = (max_seqlen + 255) / 256;
    const int64_t nwg = (int64_t)nqb * heads * nseg;
    FP_REQUIRE(nwg < (1ll << 31));
    // One workgroup per CU is resident (8 waves x 238 registers): nwg items take ceil(nwg / CUs) rounds.  When the
    // last round is at most half full and the caller gave a workspace, its items are each cut into `split` KV
    // ranges so that the round takes 1 / split of a sweep (+ a small merge) instead of a whole one.
    const int ncu = cu_count();
    int nwg_full = (int)nwg, split = 1;
    const int tail = (int)(nwg % ncu);
    if (workspace && nwg > ncu && tail > 0 && tail <= ncu / 2) {
        split = ncu / tail < 8 ? ncu / tail : 8;
        if ((int64_t)tail * split * WS_FLOATS * 4 <= workspace_bytes && ((uintptr_t)workspace & 15) == 0)
            nwg_full = (int)nwg - tail;
        else if (workspace_required)
            return (int)hipErrorInvalidValue;       // the fixed scratch cannot hold this device's last round
        else
            split = 1;
    }
    const int64_t grid = nwg_full + (int64_t)(nwg - nwg_full) * split;
    // defer-max threshold 2^8: P <= 256 before the bf16 rounding (relative precision unchanged), the O rescale
    // runs only when a row's max grows by more than 8 in the log2 domain
    if (g_attn_variant == 1) {
        auto kern = attn_w64_kernel<8>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * KBUF);
        if (e != hipSuccess) return (int)e;
        kern<<<dim3((unsigned)grid), dim3(256), 6 * KBUF, (hipStream_t)stream>>>(
            (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, q_row_stride, k_row_stride, v_row_stride,
            o_row_stride, cu_seqlens, nseg, total_tokens, heads, nqb, scale_log2e, nwg_full, split, (float*)workspace);
    } else {
        auto kern = attn_fwd_kernel<8>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KBUF);
        if (e != hipSuccess) return (int)e;
        kern<<<dim3((unsigned)grid), dim3(512), 4 * KBUF, (hipStream_t)stream>>>(
            (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, q_row_stride, k_row_stride, v_row_stride,
            o_row_stride, cu_seqlens, nseg, total_tokens, heads, nqb, scale_log2e, nwg_full, split, (float*)workspace);
    }
    FP_CHECK_LAUNCH();
    if (nwg_full < nwg)
        attn_combine_kernel<<<dim3((unsigned)((nwg - nwg_full) * 16)), dim3(256), 0, (hipStream_t)stream>>>(
            (const float*)workspace, (bf16*)o, o_row_stride, cu_seqlens, total_tokens, heads, nqb, nwg_full, split);
    FP_CHECK_LAUNCH();
    return 0;
}

extern "C" int fp_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                                int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                                const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens, int heads,
                                int head_dim, float softmax_scale, void* workspace, int64_t workspace_bytes,
                                void* stream) {
    return attn_launch(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, cu_seqlens, nseg, max_seqlen,
                       total_tokens, heads, head_dim, softmax_scale, workspace, workspace_bytes, false, stream);
}

// ---- The entry point that allocates nothing (no counterpart in the reference; for callers that cannot allocate:
// HIP-graph capture, serving loops).  Same call as fp_attn_fwd_bf16 with the split last round always on: the fp32
// partials go to a fixed array in device memory that every GPU the library runs on holds until the process ends.
// In host code `scratch` is only a stand-in for the symbol: its per-device address comes from hipGetSymbolAddress,
// once per device, and only that address is handed to kernels.
constexpr int SCRATCH_PARTIALS = 256;           // one partial per CU of an MI355X: 256 x 133 120 B = 32.5 MiB
static __device__ __align__(256) float scratch[(size_t)SCRATCH_PARTIALS * WS_FLOATS];
constexpr int SCRATCH_MAX_DEVICES = 64;
static float* g_scratch_ptr[SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return (int)hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));      // the CURRENT device's copy of the array
        if (e != hipSuccess) return (int)e;
        if (!p) return (int)hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = (float*)p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int64_t fp_attn_static_scratch_bytes(void) { return (int64_t)SCRATCH_PARTIALS * WS_FLOATS * 4; }

extern "C" int fp_attn_fwd_bf16_static(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                                       int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                                       const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens, int heads,
                                       int head_dim, float softmax_scale, void* stream) {
    float* ptr = nullptr;
    const int e = scratch_for_current_device(&ptr);
    if (e) return e;
    // every partial the kernels read is written by the same call first (attn_*_kernel stores O, m, l of each part
    // before attn_combine_kernel runs), so nothing in the array has to start at zero
    return attn_launch(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, cu_seqlens, nseg, max_seqlen,
                       total_tokens, heads, head_dim, softmax_scale, ptr, fp_attn_static_scratch_bytes(), true, stream);
}

#ifdef FP_ATTN_DIAG
// diagnostic build: per-wave cycle sums of the stamped tile loop
extern "C" int fp_attn_dbg_read(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(fp_attn_dbg), sizeof(unsigned long long) * 64);
}
#endif
