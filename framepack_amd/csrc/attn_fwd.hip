// K1: varlen flash-attention forward, bf16, head_dim 128, non-causal  (gfx950 MFMA 32x32x16, wave64)
//
// Replaces ref diffusers_helper/models/hunyuan_video_packed.py:89-112 (attn_varlen_func: the SDPA
// branch :97 is the parity target; the sageattention / flash / xformers dispatch is gone).
//
// Structure (one workgroup = 8 waves = 256 query rows of one head; each wave owns 32 query rows):
//   * Q fragments stay in registers for the whole KV sweep (32 VGPRs).
//   * K and V tiles of 64 keys are staged global -> registers -> LDS, double buffered; the global
//     loads of tile t+1 are issued before the MFMAs of tile t and written to LDS after them, so HBM/L2
//     latency hides under compute and one barrier per tile suffices.
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
#include <stdlib.h>

namespace {

constexpr int KVBLK = 64;     // keys per tile
constexpr int HD = 128;       // head dim
constexpr int KBUF = KVBLK * HD * 2;   // 16 KiB per K (or V) buffer

__device__ __forceinline__ unsigned k_lds_off(int key, int chunk) {      // chunk = 16-B piece of the 256-B row
    return (unsigned)(key * 256 + ((chunk ^ (key & 15)) << 4));
}
__device__ __forceinline__ unsigned v_lds_off(int key, int chunk) {
    return (unsigned)((chunk >> 2) * (KVBLK * 64) + key * 64 + ((chunk & 3) << 4));
}

// NW = waves per workgroup (32 query rows each); THR_LOG2 = defer-max threshold in log2 units
// (0 = rescale whenever the max grows)
template <int NW, int THR_LOG2>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, bf16* __restrict__ O,
                                                          int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                                                          const int* __restrict__ cu_seqlens, int nseg, int total,
                                                          int heads, int nqb, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* kbuf = smem;                 // [2][KBUF]
    unsigned char* vbuf = smem + 2 * KBUF;      // [2][KBUF]

    constexpr int QBLK = 32 * NW, NT = 64 * NW, CPT = 1024 / NT;   // chunks per thread per operand tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    // work item -> (segment, head, q block); consecutive items of one XCD share a head (K/V via its L2)
    const int w = xcd_remap(blockIdx.x, gridDim.x);
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
    const int nt = (nkv + KVBLK - 1) / KVBLK;

    // ---- Q fragments: B operand of S^T = K.Q^T: lane (r,hh) holds Q[q = r][d = 16ks + 8hh .. +7]
    const int qrow = min(q0 + wave * 32 + r, seg_hi - 1);
    bf16x8 qf[8];
    {
        const bf16* qp = Q + (int64_t)qrow * q_rs + head * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- staging: 1024 16-B chunks per tile per operand, CPT per thread
    const int sk0 = tid >> 4, sc = tid & 15;    // chunk ids tid + NT*i: keys sk0 + (NT/16)*i
    const bf16* kbase = K + head * HD + sc * 8;
    const bf16* vbase = V + head * HD + sc * 8;
    u32x4 kreg[CPT], vreg[CPT];
    auto stage_load = [&](int t) {
        const int kv0 = seg_lo + t * KVBLK;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int row = min(kv0 + sk0 + (NT / 16) * i, seg_hi - 1);
            kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)row * k_rs);
            vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)row * v_rs);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int key = sk0 + (NT / 16) * i;
            *reinterpret_cast<u32x4*>(kbuf + buf * KBUF + k_lds_off(key, sc)) = kreg[i];
            *reinterpret_cast<u32x4*>(vbuf + buf * KBUF + v_lds_off(key, sc)) = vreg[i];
        }
    };

    // per-lane LDS read offsets
    //   K (A operand of S^T): lane (r,hh), k-step ks: key kb*32 + r, chunk 2ks + hh
    //   V (A operand of O^T, transposed read): 16-lane group reads a 4-key x 16-d block; lane 4q+p gives
    //   the address of key row q, d columns 4p..4p+3
    const int i16 = lane & 15;
    const unsigned v_lane_off = (unsigned)((4 * hh + (i16 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2);

    f32x16 o_acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) o_acc[d][j] = 0.f;
    float m_run = -INFINITY;     // running max (log2 domain, already scaled)
    float l_run = 0.f;           // this lane's partial row sum (its half of the keys)

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage_load(t + 1);

        // ---- S^T = K.Q^T  (2 key blocks x 8 k-steps)
        f32x16 s_acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 16; ++j) s_acc[kb][j] = 0.f;
            const unsigned char* kp = kbuf + cur * KBUF;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kp + k_lds_off(kb * 32 + r, 2 * ks + hh));
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s_acc[kb], 0, 0, 0);
            }
        }

        // ---- mask the tail keys (last tile only; uniform branch)
        if (t == nt - 1 && (nkv & (KVBLK - 1))) {
            const int valid = nkv - t * KVBLK;     // keys [0, valid) of this tile exist
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int key = kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                    if (key >= valid) s_acc[kb][j] = -INFINITY;
                }
        }

        // ---- online softmax, lane-local (query row = r; the two lane halves hold disjoint keys)
        float mx = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s_acc[kb][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
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
        float rs = 0.f;
        bf16x8 pf[4];                            // B operand of O^T, k-step s = 2kb + s'
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][j], scale_log2e, -m_run));
                rs += p;
                pf[kb * 2 + (j >> 3)][j & 7] = (bf16)p;
            }
        l_run += rs;

        // ---- O^T += V^T.P^T   (4 d blocks x 4 k-steps of 16 keys)
        {
            const unsigned char* vp = vbuf + cur * KBUF + v_lane_off;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // keys 16s + 8e + 4hh + q  (e = 0,1): element j of the fragment = key 16s + 8(j>>2) + 4hh + (j&3)
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vp + d * (KVBLK * 64) + (16 * s) * 64));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vp + d * (KVBLK * 64) + (16 * s + 8) * 64));
                    union {
                        s16x4 h[2];
                        bf16x8 v;
                    } a;
                    a.h[0] = lo;
                    a.h[1] = hi;
                    o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, pf[s], o_acc[d], 0, 0, 0);
                }
            }
        }

        if (t + 1 < nt) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T / l ; lane (r,hh) holds d = 32*dblk + 8g + 4hh + (0..3) in regs 4g..4g+3
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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

// ---------------------------------------------------------------------------------------------------------
// Ping-pong variant.  Same tiles, images and fragment maps as above; what changes is WHEN things run.
// Each wave's tile step is split into a vector segment V_t (online softmax of S(t) -> P(t)) and a matrix
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

__device__ unsigned long long fp_attn_dbg[8 * 8];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FP_STAMP(var) if (PRIO >= 2) var = stamp()

template <int THR_LOG2, int PRIO>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                             const bf16* __restrict__ V, bf16* __restrict__ O,
                                                             int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                                                             const int* __restrict__ cu_seqlens, int nseg, int total,
                                                             int heads, int nqb, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* kbuf = smem;                 // [2][KBUF]
    unsigned char* vbuf = smem + 2 * KBUF;      // [2][KBUF]

    constexpr int QBLK = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 31, hh = lane >> 5;

    const int w = xcd_remap(blockIdx.x, gridDim.x);
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
    const int nt = (nkv + KVBLK - 1) / KVBLK;

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
        k_cur[i] = (unsigned)((seg_lo + sk0 + 32 * i) * k_rs * 2) + col_b;
        v_cur[i] = (unsigned)((seg_lo + sk0 + 32 * i) * v_rs * 2) + col_b;
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

    unsigned long long aLd = 0, aMx = 0, aRs = 0; unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0, T5 = 0, aV = 0, aB1 = 0, aM = 0, aW = 0, aB2 = 0;
    for (int t = 0; t < nt; ++t) {
        // ================= vector segment V_t
        FP_STAMP(T0);
        // unconditional (tile index clamped): a load under a runtime condition makes the compiler merge old
        // and new register values right behind the load, i.e. wait for it at once
        load_k();                               // K(t+2)
        load_v();                               // V(t+1)

        if (t == nt - 1 && (nkv & (KVBLK - 1))) {
            const int valid = nkv - t * KVBLK;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int key = kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                    if (key >= valid) s_acc[kb][j] = -INFINITY;
                }
        }
        unsigned long long Ta = 0, Tb = 0, Tc = 0;
        FP_STAMP(Ta);
        float mx = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s_acc[kb][j]);
        mx = fmaxf(mx, other_half(mx)) * scale_log2e;
        if (PRIO >= 2) asm volatile("" ::"v"(mx));
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
        if (PRIO >= 2) asm volatile("" ::"v"(o_acc[0]), "v"(o_acc[3]));
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

        if (PRIO >= 2) asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(l_run));
        FP_STAMP(T1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        FP_STAMP(T2);

        // ================= matrix segment M_t
        // Eight groups of four MFMAs: O^T d-blocks 0..3 (V^T fragments: 8 transposed reads each), then
        // S^T(t+1) as (key block, k-step quad) x 4 (K fragments: 4 ds_read_b128 each).  The LDS reads of
        // group g+1 are issued before the MFMAs of group g (two fragment sets), so that only the first
        // group's LDS latency is exposed; the scheduling barriers keep hipcc from undoing that order.
        if (PRIO == 1 || PRIO == 2) __builtin_amdgcn_s_setprio(1);
        {
            const unsigned char* vp = vbuf + (t & 1) * KBUF + v_lane_off;
            const unsigned char* kp = kbuf + ((t + 1) & 1) * KBUF;
            // S^T(t+1) is computed after the last tile too (from a stale K buffer, result unused): keeping the
            // segment branch-free lets the LDS-read counts of consecutive groups chain exactly.
            bf16x8 fa[2][4];
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
            rd_v(fa[0], 0);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (d < 3) rd_v(fa[(d + 1) & 1], d + 1);
                else rd_k(fa[0], 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[d & 1][s], pf[s], o_acc[d], 0, 0, 0);
                if (d == 1) {
                    // the tiles loaded at the start of V_t go to the LDS buffers nobody reads in this segment
                    if (t + 2 < nt) write_k(t & 1);
                    if (t + 1 < nt) write_v((t + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {
#pragma unroll
                for (int g = 0; g < 4; ++g) {        // g = 2 kb + kq
                    if (g < 3) rd_k(fa[(g + 1) & 1], (g + 1) >> 1, (g + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((g & 1) == 0) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) s_acc[g >> 1][j] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        s_acc[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][i], qf[4 * (g & 1) + i],
                                                                                s_acc[g >> 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (PRIO == 1 || PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (PRIO >= 2) asm volatile("" ::"v"(s_acc[0]), "v"(s_acc[1]), "v"(o_acc[0]), "v"(o_acc[3]));
        FP_STAMP(T3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FP_STAMP(T4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        FP_STAMP(T5);
        if (PRIO >= 2) {
            aV += T1 - Tc; aLd += Ta - T0; aMx += Tb - Ta; aRs += Tc - Tb; aB1 += T2 - T1; aM += T3 - T2; aW += T4 - T3; aB2 += T5 - T4;
        }
    }
    if (PRIO >= 2 && blockIdx.x == 8 && lane == 0) {
        fp_attn_dbg[wave * 8 + 0] = aV; fp_attn_dbg[wave * 8 + 1] = aB1; fp_attn_dbg[wave * 8 + 2] = aM;
        fp_attn_dbg[wave * 8 + 3] = aW; fp_attn_dbg[wave * 8 + 4] = aB2; fp_attn_dbg[wave * 8 + 5] = nt; fp_attn_dbg[wave * 8 + 6] = aLd; fp_attn_dbg[wave * 8 + 7] = (aMx << 32) | aRs;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    const float l_tot = l_run + other_half(l_run);
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

}  // namespace

// q, k, v, o: bf16 [total_tokens, heads, 128] with arbitrary row strides (elements) and head stride 128.
// cu_seqlens: device int32 [nseg+1] or NULL for a single segment [0, total_tokens).
// o may alias q (each workgroup reads its own Q rows before it writes them).
extern "C" int fp_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                                int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                                const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens, int heads,
                                int head_dim, float softmax_scale, void* stream) {
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
    int variant = 0;
    if (const char* e = getenv("FP_ATTN_VARIANT")) variant = atoi(e);
#define FP_LAUNCH_ATTN(NW_, THR_)                                                                                  \
    do {                                                                                                           \
        const int nqb = (max_seqlen + 32 * NW_ - 1) / (32 * NW_);                                                  \
        const int64_t nwg = (int64_t)nqb * heads * nseg;                                                           \
        FP_REQUIRE(nwg < (1ll << 31));                                                                             \
        hipError_t e_ = hipFuncSetAttribute((const void*)attn_fwd_kernel<NW_, THR_>,                               \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KBUF);                 \
        if (e_ != hipSuccess) return (int)e_;                                                                      \
        attn_fwd_kernel<NW_, THR_><<<dim3((unsigned)nwg), dim3(64 * NW_), 4 * KBUF, (hipStream_t)stream>>>(        \
            (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, q_row_stride, k_row_stride, v_row_stride,    \
            o_row_stride, cu_seqlens, nseg, total_tokens, heads, nqb, scale_log2e);                                \
    } while (0)
#define FP_LAUNCH_ATTN_PP(THR_, PRIO_)                                                                              \
    do {                                                                                                           \
        const int nqb = (max_seqlen + 255) / 256;                                                                  \
        const int64_t nwg = (int64_t)nqb * heads * nseg;                                                           \
        FP_REQUIRE(nwg < (1ll << 31));                                                                             \
        hipError_t e_ = hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<THR_, PRIO_>,                          \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KBUF);                 \
        if (e_ != hipSuccess) return (int)e_;                                                                      \
        attn_fwd_pp_kernel<THR_, PRIO_><<<dim3((unsigned)nwg), dim3(512), 4 * KBUF, (hipStream_t)stream>>>(        \
            (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, q_row_stride, k_row_stride, v_row_stride,    \
            o_row_stride, cu_seqlens, nseg, total_tokens, heads, nqb, scale_log2e);                                \
    } while (0)
    switch (variant) {
        case 4: FP_LAUNCH_ATTN_PP(0, 0); break;
        case 5: FP_LAUNCH_ATTN_PP(0, 1); break;
        case 6: FP_LAUNCH_ATTN_PP(8, 1); break;
        case 7: FP_LAUNCH_ATTN_PP(0, 2); break;
        case 8: FP_LAUNCH_ATTN_PP(0, 3); break;
        case 1: FP_LAUNCH_ATTN(4, 0); break;
        case 2: FP_LAUNCH_ATTN(8, 8); break;
        case 3: FP_LAUNCH_ATTN(4, 8); break;
        default: FP_LAUNCH_ATTN(8, 0); break;
    }
    FP_CHECK_LAUNCH();
    return 0;
}

// diagnostic: per-wave cycle sums of the stamped build (variant 7)
extern "C" int fp_attn_dbg_read(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(fp_attn_dbg), sizeof(unsigned long long) * 64);
}
