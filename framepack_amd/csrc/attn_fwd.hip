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

// Bytes of device workspace with which fp_attn_fwd_bf16 can split the last round's work items (any shape).
extern "C" int64_t fp_attn_workspace_bytes(void) { return (int64_t)cu_count() * WS_FLOATS * 4; }

// q, k, v, o: bf16 [total_tokens, heads, 128] with arbitrary row strides (elements) and head stride 128.
// cu_seqlens: device int32 [nseg+1] or NULL for a single segment [0, total_tokens).
// o may alias q (each workgroup reads its own Q rows before it writes them).
extern "C" int fp_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                                int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                                const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens, int heads,
                                int head_dim, float softmax_scale, void* workspace, int64_t workspace_bytes,
                                void* stream) {
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
        else
            split = 1;
    }
    const int64_t grid = nwg_full + (int64_t)(nwg - nwg_full) * split;
    // defer-max threshold 2^8: P <= 256 before the bf16 rounding (relative precision unchanged), the O rescale
    // runs only when a row's max grows by more than 8 in the log2 domain
    auto kern = attn_fwd_kernel<8>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KBUF);
    if (e != hipSuccess) return (int)e;
    kern<<<dim3((unsigned)grid), dim3(512), 4 * KBUF, (hipStream_t)stream>>>(
        (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, q_row_stride, k_row_stride, v_row_stride,
        o_row_stride, cu_seqlens, nseg, total_tokens, heads, nqb, scale_log2e, nwg_full, split, (float*)workspace);
    FP_CHECK_LAUNCH();
    if (nwg_full < nwg)
        attn_combine_kernel<<<dim3((unsigned)((nwg - nwg_full) * 16)), dim3(256), 0, (hipStream_t)stream>>>(
            (const float*)workspace, (bf16*)o, o_row_stride, cu_seqlens, total_tokens, heads, nqb, nwg_full, split);
    FP_CHECK_LAUNCH();
    return 0;
}

#ifdef FP_ATTN_DIAG
// diagnostic build: per-wave cycle sums of the stamped tile loop
extern "C" int fp_attn_dbg_read(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(fp_attn_dbg), sizeof(unsigned long long) * 64);
}
#endif
