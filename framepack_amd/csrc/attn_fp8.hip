// K1, fp8 variant: flash-attention forward with e4m3 Q, K, P and V on the block-scaled MFMA 32x32x64 (unit block
// scales), head_dim 128, one segment, non-causal.  Opt-in: e4m3 has 3 mantissa bits, so the result differs from
// the bf16 kernel's by a few per cent (tests/test_attn_fp8_gpu.py states the measured distance) - outside the
// 1e-2 parity bar of the SDPA path (ref diffusers_helper/models/hunyuan_video_packed.py:97), in the role the
// reference gives its quantised SageAttention branch (:92-94).
//
//   quantisation (inside fp_attn_fwd_fp8, into the caller's workspace)   q, k, v bf16 [S, heads*128] (row-strided)  ->
//        q8, k8  e4m3 [S, heads*128], ONE scale per head (amax over all rows and the head's 128 columns / 448):
//                q and k are RMS-normalised, so a head's rows share their magnitude; the two scales fold into the
//                softmax scale and cost nothing in the kernel.
//        vt8     e4m3 [heads][tiles of 64 keys][128 d][64 key slots]: V TRANSPOSED, one scale per (head, channel)
//                (applied to the output row at the end).  The 64 key slots of a tile are in the order in which the
//                S^T accumulator of the kernel holds its keys (slot p: half hh = p / 32, i = p % 32, key =
//                32 (i / 16) + (i & 3) + 8 ((i % 16) / 4) + 4 hh), so a lane's MFMA operand is 32 contiguous bytes
//                and no transposing LDS read is needed.  Keys past S are zero.
//        scales  fp32 [2 heads + 128 heads]: sq[h], sk[h], sv[h][d].
//   then the sweep         O bf16 = softmax(Q K^T * softmax_scale) V from the quantised operands.
//
// Kernel structure = the first-generation bf16 kernel of attn_fwd.hip (8 waves x 32 query rows, two waves per SIMD
// alternating a vector and a matrix segment, K two tiles and V one tile ahead in LDS), with
//   * S^T(kb) = K8 . Q8^T: 2 MFMAs of K = 64 per 32-key block (16 in bf16), Q fragments 16 VGPRs;
//   * O^T(d block) += V8^T . P8^T: ONE MFMA per d block and tile (4 in bf16); P is packed to e4m3 straight from
//     the accumulator registers (P <= 2^8 by the defer-max threshold, inside e4m3's range);
//   * K and V tiles are 8 KiB each: half the LDS and global traffic;
//   * (round 2) the S accumulators hold the softmax exponent itself: sq is chosen so that softmax_scale log2(e) sq sk is a
//     power of two, which the S MFMAs apply as their block scale, and -m (stale until a tile's row sum passes 2^8) is the
//     C operand of a chain's first MFMA: the vector segment is exp2, add, pack - no fma, no max, no select (82 vector
//     instructions per tile and wave, was ~180).  The sweep is then bound by LDS reads: 8 waves x 32 rows read every tile
//     8 times, 128 KiB per tile and CU (a 4-wave, 64-row sweep like attn_v3_kernel would halve that; not built).
#include "fp_common.h"

// attn_fwd.hip: merge of the split tail items (same partial layout: O[256][128] unnormalised, m[256], l[256], fp32)
int fpi_attn_cu_count();
int fpi_attn_combine(const float* ws, void* o, int64_t o_rs, const int* cu_seqlens, int total, int heads, int nqb,
                     int nwg_full, int split, int ntail, hipStream_t st);

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int KVBLK = 64, HD = 128, QBLK = 256;
constexpr int TILE_BYTES = KVBLK * HD;          // 8 KiB: one K or V tile in e4m3
constexpr int WS_FLOATS = 256 * HD + 2 * 256;   // one partial sweep, as in attn_fwd.hip
constexpr float FP8_MAX = 448.f;

__host__ __device__ __forceinline__ int slot_key(int p) {
    const int hh = p >> 5, i = p & 31, j = i & 15;
    return 32 * (i >> 4) + (j & 3) + 8 * (j >> 2) + 4 * hh;
}

// ---- quantisation -----------------------------------------------------------------------------------------------
// column-wise |max| of q, k and v over all rows, in two deterministic stages without atomics (606 row blocks
// hammering 9216 addresses with atomic max took longer than reading the data): partial[row block][3][C], then
// amax[3][C].  Block = 32 rows x 256 chunks of 8 columns (grid.y walks the chunks of the three tensors): one
// chunk per thread, rows read as 4 KiB runs per wave.
constexpr int AMAX_ROWS = 32;
__global__ __launch_bounds__(256) void absmax_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                     const bf16* __restrict__ v, int64_t q_rs, int64_t k_rs,
                                                     int64_t v_rs, int S, int C, float* __restrict__ partial) {
    const int r0 = blockIdx.x * AMAX_ROWS, r1 = min(r0 + AMAX_ROWS, S);
    const int nch = C / 8;
    const int item = blockIdx.y * 256 + threadIdx.x;
    if (item >= 3 * nch) return;
    const int which = item / nch, c = item - which * nch;
    const bf16* base = (which == 0 ? q : (which == 1 ? k : v)) + c * 8;
    const int64_t rs = which == 0 ? q_rs : (which == 1 ? k_rs : v_rs);
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(base + (int64_t)r * rs);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], fabsf((float)x[j]));
    }
    float* out = partial + (int64_t)blockIdx.x * 3 * C + which * C + c * 8;
    *reinterpret_cast<f32x4*>(out) = f32x4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<f32x4*>(out + 4) = f32x4{m[4], m[5], m[6], m[7]};
}

__global__ __launch_bounds__(256) void absmax_reduce_kernel(const float* __restrict__ partial, int nblk, int n,
                                                            float* __restrict__ amax) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m = 0.f;
#pragma unroll 8
    for (int b = 0; b < nblk; ++b) m = fmaxf(m, partial[(int64_t)b * n + i]);
    amax[i] = m;
}

// scales[0..H) = sq, [H..2H) = sk, [2H..2H+128H) = sv  (amax / 448; 1 where the amax is 0).  sq is then raised (by a
// factor below 2: e4m3 keeps its relative precision) so that c = softmax_scale * log2(e) * sq * sk is a power of two: the
// sweep applies c as the MFMA's block scale and its accumulator holds the exponent of the softmax directly.
__global__ __launch_bounds__(128) void scales_kernel(const float* __restrict__ amax, int heads, float scale_log2e,
                                                     float* __restrict__ scales) {
    __shared__ float red[2][2];
    const int h = blockIdx.x, d = threadIdx.x, C = heads * HD;
    float mq = amax[h * HD + d], mk = amax[C + h * HD + d];
    const float mv = amax[2 * C + h * HD + d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mq = fmaxf(mq, __shfl_xor(mq, o, 64));
        mk = fmaxf(mk, __shfl_xor(mk, o, 64));
    }
    if ((d & 63) == 0) {
        red[0][d >> 6] = mq;
        red[1][d >> 6] = mk;
    }
    __syncthreads();
    if (d == 0) {
        mq = fmaxf(red[0][0], red[0][1]);
        mk = fmaxf(red[1][0], red[1][1]);
        const float sq = mq > 0.f ? mq * (1.f / FP8_MAX) : 1.f, sk = mk > 0.f ? mk * (1.f / FP8_MAX) : 1.f;
        const float c = scale_log2e * sq * sk;
        const float c2 = exp2f(ceilf(log2f(c)));                     // the next power of two (c itself if it is one)
        scales[h] = sq * (c2 / c);
        scales[heads + h] = sk;
    }
    scales[2 * heads + h * HD + d] = mv > 0.f ? mv * (1.f / FP8_MAX) : 1.f;
}

__device__ __forceinline__ u32x2 quant8(const bf16x8 v, float inv) {
    u32x2 o;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[0] * inv, (float)v[1] * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[2] * inv, (float)v[3] * inv, w, true);
    o[0] = (unsigned)w;
    w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[4] * inv, (float)v[5] * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[6] * inv, (float)v[7] * inv, w, true);
    o[1] = (unsigned)w;
    return o;
}

// q8 / k8: one thread per 8 columns of a row
__global__ __launch_bounds__(256) void quant_qk_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                       int64_t q_rs, int64_t k_rs, int S, int heads,
                                                       const float* __restrict__ scales,
                                                       unsigned char* __restrict__ q8, unsigned char* __restrict__ k8) {
    const int C = heads * HD, nch = C / 8;
    const int64_t n = (int64_t)S * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / nch), c = (int)(i - (int64_t)row * nch);
        const int h = c / (HD / 8);
        const bf16x8 xq = *reinterpret_cast<const bf16x8*>(q + (int64_t)row * q_rs + c * 8);
        const bf16x8 xk = *reinterpret_cast<const bf16x8*>(k + (int64_t)row * k_rs + c * 8);
        *reinterpret_cast<u32x2*>(q8 + (int64_t)row * C + c * 8) = quant8(xq, 1.f / scales[h]);
        *reinterpret_cast<u32x2*>(k8 + (int64_t)row * C + c * 8) = quant8(xk, 1.f / scales[heads + h]);
    }
}

// vt8: block = (segment * tps + tile of 64 keys, head); through an LDS tile [64 keys][128 d] of e4m3.  Every segment
// has its own grid of `tps` tiles starting at its first row (tile t of segment s covers rows seg_lo + 64 t ..), so
// the sweep of a segment reads whole tiles whatever the segment's offset in the packed rows.
__global__ __launch_bounds__(256) void quant_vt_kernel(const bf16* __restrict__ v, int64_t v_rs, int S, int heads,
                                                       const int* __restrict__ cu_seqlens, int nseg, int tps,
                                                       const float* __restrict__ scales,
                                                       unsigned char* __restrict__ vt8) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[KVBLK][HD + 16];
    const int seg = blockIdx.x / tps, t = blockIdx.x % tps, h = blockIdx.y, tid = threadIdx.x;
    int seg_lo = 0, seg_hi = S;
    if (cu_seqlens) {
        seg_lo = cu_seqlens[seg];
        seg_hi = cu_seqlens[seg + 1];
    }
    if (seg_lo + t * KVBLK >= seg_hi) return;          // (uniform) no key of this segment in the tile: never read
    const float* sv = scales + 2 * heads + h * HD;
    for (int item = tid; item < KVBLK * (HD / 8); item += 256) {
        const int key = item / (HD / 8), c = item % (HD / 8);
        const int row = seg_lo + t * KVBLK + key;
        u32x2 o = {0u, 0u};
        if (row < seg_hi) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(v + (int64_t)row * v_rs + h * HD + c * 8);
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[0] / sv[c * 8 + 0], (float)x[1] / sv[c * 8 + 1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[2] / sv[c * 8 + 2], (float)x[3] / sv[c * 8 + 3], w, true);
            o[0] = (unsigned)w;
            w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[4] / sv[c * 8 + 4], (float)x[5] / sv[c * 8 + 5], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)x[6] / sv[c * 8 + 6], (float)x[7] / sv[c * 8 + 7], w, true);
            o[1] = (unsigned)w;
        }
        *reinterpret_cast<u32x2*>(&tile[key][c * 8]) = o;
    }
    __syncthreads();
    unsigned char* out = vt8 + (((int64_t)h * nseg + seg) * tps + t) * TILE_BYTES;
    for (int item = tid; item < HD * (KVBLK / 4); item += 256) {          // 4 slots (one dword) per item
        const int d = item / (KVBLK / 4), p0 = (item % (KVBLK / 4)) * 4;
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) w |= (unsigned)tile[slot_key(p0 + b)][d] << (8 * b);
        *reinterpret_cast<unsigned*>(out + d * KVBLK + p0) = w;
    }
}

// ---- attention --------------------------------------------------------------------------------------------------
// K image: [key][128 B], 16-B chunk index XOR ((key >> 1) & 7);  V^T image: [d][64 B], chunk XOR ((d >> 2) & 3):
// both conflict-free for the ds_read_b128 pairs below.
__device__ __forceinline__ unsigned k_off(int key, int chunk) {
    return (unsigned)(key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4));
}
__device__ __forceinline__ unsigned v_off(int d, int chunk) {
    return (unsigned)(d * 64 + ((chunk ^ ((d >> 2) & 3)) << 4));
}
__device__ __forceinline__ i32x8 cat(const i32x4 a, const i32x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x16 mfma8(const i32x8 a, const i32x8 b, const f32x16 c) {
    // cbsz = blgp = 0: both operands e4m3; scale operands 0 = unit block scales
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}
// the same with block scale 2^(sa - 127) on the product (every lane's K block: the byte op_sel 0 picks), C added unscaled
__device__ __forceinline__ f32x16 mfma8s(const i32x8 a, const i32x8 b, const f32x16 c, int sa) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, 127);
}
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

template <int THR_LOG2>
__global__ __launch_bounds__(512, 2) void attn_fp8_kernel(const unsigned char* __restrict__ Q8,
                                                          const unsigned char* __restrict__ K8,
                                                          const unsigned char* __restrict__ VT8,
                                                          const float* __restrict__ scales, bf16* __restrict__ O,
                                                          int64_t o_rs, const int* __restrict__ cu_seqlens,
                                                          int nseg, int tps, int S, int heads, int nqb,
                                                          float scale_log2e, int nwg_full, int split,
                                                          float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* kbuf = smem;                       // [2][8 KiB]
    unsigned char* vbuf = smem + 2 * TILE_BYTES;      // [2][8 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 31, hh = lane >> 5;

    // items [0, nwg_full) are whole sweeps; the items of the last, partial round are cut into `split` KV ranges that
    // write partial (O, m, l) for the merge kernel of attn_fwd.hip (same scheme as the bf16 kernel)
    int w, part = 0, nparts = 1;
    if ((int)blockIdx.x < nwg_full) {
        w = xcd_remap(blockIdx.x, nwg_full);
    } else {
        const int si = blockIdx.x - nwg_full;
        w = nwg_full + si / split;
        part = si % split;
        nparts = split;
    }
    // work item = (segment, head, block of 256 query rows of the segment); segment s covers rows cu[s] .. cu[s+1]
    const int qb = w % nqb, head = (w / nqb) % heads, seg = w / (nqb * heads);
    int seg_lo = 0, seg_hi = S;
    if (cu_seqlens) {
        seg_lo = cu_seqlens[seg];
        seg_hi = cu_seqlens[seg + 1];
    }
    const int q0 = seg_lo + qb * QBLK;
    if (q0 >= seg_hi) return;                                        // (uniform) no query row of the segment here
    const int C = heads * HD;
    const int nkv = seg_hi - seg_lo;
    const int ntiles = min((nkv + KVBLK - 1) / KVBLK, tps);          // (a segment longer than max_seqlen stays in its grid)
    const int t0 = (int)((int64_t)ntiles * part / nparts);           // this block's KV tiles: [t0, t0 + nt)
    const int nt = (int)((int64_t)ntiles * (part + 1) / nparts) - t0;
    float* wsp = ws + (int64_t)((w - nwg_full) * split + part) * WS_FLOATS;
    if (nt == 0) {                                                   // (uniform) more parts than tiles: a neutral partial
        for (int i = tid; i < 256; i += 512) {
            wsp[256 * HD + i] = -INFINITY;
            wsp[256 * HD + 256 + i] = 0.f;
        }
        return;
    }
    // softmax scale x log2(e) x sq x sk = 2^e exactly (scales_kernel): applied by the S MFMAs as their block scale
    const int s_exp = 127 + (int)rintf(log2f(scale_log2e * scales[head] * scales[heads + head]));

    // Q fragments: d = 64 ks + 32 hh .. + 32 of query row r
    i32x8 qf[2];
    {
        const int qrow = min(q0 + wave * 32 + r, seg_hi - 1);
        const unsigned char* qp = Q8 + (int64_t)qrow * C + head * HD + hh * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = cat(*reinterpret_cast<const i32x4*>(qp + ks * 64), *reinterpret_cast<const i32x4*>(qp + ks * 64 + 16));
    }

    // staging: one 16-B piece of K and one of V^T per thread and tile
    const int skey = tid >> 3, sch = tid & 7;
    const unsigned char* kp_g = K8 + head * HD + sch * 16;
    const unsigned char* vp_g = VT8 + ((int64_t)head * nseg + seg) * tps * TILE_BYTES + tid * 16;
    const unsigned k_dst = k_off(skey, sch), v_dst = v_off(tid >> 2, tid & 3);
    int kt = 0, vt = 0;                                // next tile to load
    i32x4 kreg, vreg;
    auto load_k = [&]() {
        const int row = min(seg_lo + (t0 + min(kt, nt - 1)) * KVBLK + skey, seg_hi - 1);
        kreg = *reinterpret_cast<const i32x4*>(kp_g + (int64_t)row * C);
        ++kt;
    };
    auto load_v = [&]() {
        vreg = *reinterpret_cast<const i32x4*>(vp_g + (int64_t)(t0 + min(vt, nt - 1)) * TILE_BYTES);
        ++vt;
    };
    auto write_k = [&](int buf) { *reinterpret_cast<i32x4*>(kbuf + buf * TILE_BYTES + k_dst) = kreg; };
    auto write_v = [&](int buf) { *reinterpret_cast<i32x4*>(vbuf + buf * TILE_BYTES + v_dst) = vreg; };

    auto rd_k = [&](const unsigned char* kp, int kb, int ks) {
        const int key = kb * 32 + r;
        return cat(*reinterpret_cast<const i32x4*>(kp + k_off(key, 4 * ks + 2 * hh)),
                   *reinterpret_cast<const i32x4*>(kp + k_off(key, 4 * ks + 2 * hh + 1)));
    };
    auto rd_v = [&](const unsigned char* vp, int db) {
        const int d = db * 32 + r;
        return cat(*reinterpret_cast<const i32x4*>(vp + v_off(d, 2 * hh)),
                   *reinterpret_cast<const i32x4*>(vp + v_off(d, 2 * hh + 1)));
    };

    f32x16 o_acc[4], s_acc[2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) o_acc[d][j] = 0.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    i32x8 pf;
    // The S accumulators hold the softmax exponent itself: S(t) = 2^e K.Q^T - m with -m as the C operand of a chain's
    // first MFMA (negm: the value in all 16 registers; the lane is the query row).  m may be stale: it is moved only
    // when a tile's P would leave e4m3's range - seen on the lane's partial row sum, limit 2^THR; the first tile's
    // limit is -1, so it always takes that path and sets m to its true maximum.
    float m_run = 0.f, l_run = 0.f, lim_cur = -1.f;
    f32x16 negm = zero;

    // ---- prologue: K(0), V(0), K(1) into LDS; S(0)
    load_k();
    load_v();
    write_k(0);
    write_v(0);
    load_k();
    write_k(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        s_acc[kb] = mfma8s(rd_k(kbuf, kb, 0), qf[0], negm, s_exp);
        s_acc[kb] = mfma8s(rd_k(kbuf, kb, 1), qf[1], s_acc[kb], s_exp);
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nt; ++t) {
        // ================= vector segment: online softmax of S(t) -> P(t) in e4m3
        load_k();                               // K(t+2)  (clamped repeats past the end)
        load_v();                               // V(t+1)
        if (t0 + t == ntiles - 1 && (nkv & (KVBLK - 1))) {
            int valid = nkv - (t0 + t) * KVBLK - 4 * hh;
            asm volatile("" : "+v"(valid));     // (a real branch: if-converted, its 32 selects ran in every tile)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (kb * 32 + (j & 3) + 8 * (j >> 2) >= valid) s_acc[kb][j] = -INFINITY;
            asm volatile("" : "+v"(s_acc[0]), "+v"(s_acc[1]));      // (the masked values are made in here)
        }
        float rs = 0.f;
        auto probs = [&]() __attribute__((always_inline)) {
            rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float p[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p[j] = __builtin_amdgcn_exp2f(s_acc[kb][4 * g + j]);
                        rs += p[j];
                    }
                    int wd = 0;
                    wd = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], wd, false);
                    wd = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], wd, true);
                    pf[kb * 4 + g] = wd;
                }
        };
        probs();
        if (__builtin_amdgcn_ballot_w64(!(rs <= lim_cur)) != 0) {      // (uniform) the max grew past 2^THR, or t == 0
            const bool first = t == 0;
            lim_cur = __builtin_ldexpf(1.f, THR_LOG2);
            float mx = s_acc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s_acc[kb][j]);
            mx = fmaxf(mx, other_half(mx));
            // the accumulators hold s - m: the row's max has grown by mx (if positive); rows that did not grow keep m
            const float delta = first ? mx : fmaxf(mx, 0.f);
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int j = 0; j < 16; ++j) negm[j] = -m_run;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) s_acc[kb][j] -= delta;
            if (!first) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o_acc[d][j] *= alpha;
            }
            probs();
        }
        l_run += rs;

        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        // ================= matrix segment: O^T += V(t)^T . P(t), then S(t+1) = K(t+1) . Q^T
        __builtin_amdgcn_s_setprio(1);
        {
            const unsigned char* vp = vbuf + (t & 1) * TILE_BYTES;
            const unsigned char* kp = kbuf + ((t + 1) & 1) * TILE_BYTES;
            i32x8 fa[3];
            auto rd = [&](int g) { return g < 4 ? rd_v(vp, g) : rd_k(kp, (g - 4) >> 1, (g - 4) & 1); };
            fa[0] = rd(0);
            fa[1] = rd(1);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 2 < 8) fa[(g + 2) % 3] = rd(g + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (g < 4) {
                    o_acc[g] = mfma8(fa[g % 3], pf, o_acc[g]);
                } else {
                    const int kb = (g - 4) >> 1, ks = (g - 4) & 1;
                    s_acc[kb] = mfma8s(fa[g % 3], qf[ks], ks == 0 ? negm : s_acc[kb], s_exp);
                }
                if (g == 2) {
                    if (t + 2 < nt) write_k(t & 1);
                    if (t + 1 < nt) write_v((t + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    const float l_tot = l_run + other_half(l_run);
    if (nparts > 1) {                           // partial sweep: unnormalised O (fp32, v scales applied), max and sum
        const int row = wave * 32 + r;
        float* op = wsp + row * HD + 4 * hh;
        const float* sv = scales + 2 * heads + head * HD + 4 * hh;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sv + d * 32 + g * 8);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o_acc[d][4 * g + j] * s4[j];
                *reinterpret_cast<f32x4*>(op + d * 32 + g * 8) = v;
            }
        if (hh == 0) {
            wsp[256 * HD + row] = m_run;
            wsp[256 * HD + 256 + row] = l_tot;
        }
        return;
    }
    const float inv = 1.f / l_tot;
    const int qrow = q0 + wave * 32 + r;
    if (qrow < seg_hi) {
        bf16* op = O + (int64_t)qrow * o_rs + head * HD + 4 * hh;
        const float* sv = scales + 2 * heads + head * HD + 4 * hh;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sv + d * 32 + g * 8);
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (bf16)(o_acc[d][4 * g + j] * inv * s4[j]);
                *reinterpret_cast<bf16x4*>(op + d * 32 + g * 8) = v;
            }
    }
}

}  // namespace

// Bytes of the quantised operands: q8 + k8 (S*heads*128 each), vt8 (heads * nseg * ceil(max_seqlen/64) * 8192: every
// segment has its own tile grid), scales (fp32: 2*heads + 128*heads), the column amax (3*heads*128 fp32) and its
// per-row-block partials, the partials of a split last round.  Layout inside the buffer, in this order, every part
// 256-byte aligned.
static int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }
extern "C" int64_t fp_attn_fp8_varlen_workspace_bytes(int total_tokens, int heads, int nseg, int max_seqlen) {
    const int64_t C = (int64_t)heads * HD, tps = (max_seqlen + KVBLK - 1) / KVBLK;
    const int64_t nblk = (total_tokens + AMAX_ROWS - 1) / AMAX_ROWS;
    return 2 * align256((int64_t)total_tokens * C) + align256(heads * nseg * tps * TILE_BYTES) +
           align256((2 * heads + C) * 4) + align256(3 * C * 4) + align256(nblk * 3 * C * 4) +
           align256((int64_t)fpi_attn_cu_count() * WS_FLOATS * 4);
}
extern "C" int64_t fp_attn_fp8_workspace_bytes(int total_tokens, int heads) {
    return fp_attn_fp8_varlen_workspace_bytes(total_tokens, heads, 1, total_tokens);
}

extern "C" int fp_attn_fwd_fp8_varlen(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                                      int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                                      const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens, int heads,
                                      int head_dim, float softmax_scale, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
    FP_REQUIRE(q && k && v && o && workspace && head_dim == HD && heads > 0 && heads <= 4096 && total_tokens >= 0);
    FP_REQUIRE(nseg >= 1 && (cu_seqlens || nseg == 1) && max_seqlen >= 0 && max_seqlen <= total_tokens);
    FP_REQUIRE(q_row_stride % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0 && o_row_stride % 4 == 0);
    FP_REQUIRE(q_row_stride >= (int64_t)heads * HD && k_row_stride >= (int64_t)heads * HD &&
               v_row_stride >= (int64_t)heads * HD && o_row_stride >= (int64_t)heads * HD);
    FP_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 &&
               ((uintptr_t)o & 7) == 0 && ((uintptr_t)workspace & 255) == 0);
    FP_REQUIRE(workspace_bytes >= fp_attn_fp8_varlen_workspace_bytes(total_tokens, heads, nseg, max_seqlen));
    if (total_tokens == 0 || max_seqlen == 0) return 0;
    const int S = total_tokens, C = heads * HD, tps = (max_seqlen + KVBLK - 1) / KVBLK;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* p = (unsigned char*)workspace;
    unsigned char* q8 = p;
    p += align256((int64_t)S * C);
    unsigned char* k8 = p;
    p += align256((int64_t)S * C);
    unsigned char* vt8 = p;
    p += align256((int64_t)heads * nseg * tps * TILE_BYTES);
    float* scales = (float*)p;
    p += align256((2 * heads + C) * 4);
    float* amax = (float*)p;
    p += align256(3 * C * 4);
    float* partial = (float*)p;
    const int nblk = (S + AMAX_ROWS - 1) / AMAX_ROWS;
    p += align256((int64_t)nblk * 3 * C * 4);
    float* tail_ws = (float*)p;
    absmax_kernel<<<dim3((unsigned)nblk, (unsigned)((3 * C / 8 + 255) / 256)), dim3(256), 0, st>>>(
        (const bf16*)q, (const bf16*)k, (const bf16*)v, q_row_stride, k_row_stride, v_row_stride, S, C, partial);
    FP_CHECK_LAUNCH();
    absmax_reduce_kernel<<<dim3((unsigned)((3 * C + 255) / 256)), dim3(256), 0, st>>>(partial, nblk, 3 * C, amax);
    FP_CHECK_LAUNCH();
    scales_kernel<<<dim3((unsigned)heads), dim3(128), 0, st>>>(amax, heads, softmax_scale * 1.4426950408889634f, scales);
    FP_CHECK_LAUNCH();
    quant_qk_kernel<<<dim3(2048), dim3(256), 0, st>>>((const bf16*)q, (const bf16*)k, q_row_stride, k_row_stride, S, heads,
                                                      scales, q8, k8);
    FP_CHECK_LAUNCH();
    FP_REQUIRE((int64_t)nseg * tps < (1ll << 31));
    quant_vt_kernel<<<dim3((unsigned)(nseg * tps), (unsigned)heads), dim3(256), 0, st>>>(
        (const bf16*)v, v_row_stride, S, heads, cu_seqlens, nseg, tps, scales, vt8);
    FP_CHECK_LAUNCH();
    const int nqb = (max_seqlen + QBLK - 1) / QBLK;
    const int64_t nwg = (int64_t)nqb * heads * nseg;
    FP_REQUIRE(nwg < (1ll << 31));
    // one workgroup per CU is resident: when the last round is at most half full its items are cut into KV ranges
    const int ncu = fpi_attn_cu_count();
    int nwg_full = (int)nwg, split = 1;
    const int tail = (int)(nwg % ncu);
    if (nwg > ncu && tail > 0 && tail <= ncu / 2) {
        split = ncu / tail < 8 ? ncu / tail : 8;
        nwg_full = (int)nwg - tail;
    }
    const int64_t grid = nwg_full + (int64_t)(nwg - nwg_full) * split;
    auto kern = attn_fp8_kernel<8>;
    kern<<<dim3((unsigned)grid), dim3(512), 4 * TILE_BYTES, st>>>(q8, k8, vt8, scales, (bf16*)o, o_row_stride, cu_seqlens, nseg,
                                                                tps, S, heads, nqb, softmax_scale * 1.4426950408889634f,
                                                                nwg_full, split, tail_ws);
    FP_CHECK_LAUNCH();
    if (nwg_full < nwg)
        return fpi_attn_combine(tail_ws, o, o_row_stride, cu_seqlens, S, heads, nqb, nwg_full, split, (int)(nwg - nwg_full), st);
    return 0;
}

extern "C" int fp_attn_fwd_fp8(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                               int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int total_tokens,
                               int heads, int head_dim, float softmax_scale, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return fp_attn_fwd_fp8_varlen(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, nullptr, 1,
                                  total_tokens, total_tokens, heads, head_dim, softmax_scale, workspace, workspace_bytes,
                                  stream);
}
