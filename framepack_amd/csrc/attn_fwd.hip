// K1: varlen flash-attention forward, bf16, head_dim 128, non-causal  (gfx950 MFMA 32x32x16, wave64)
//
// Replaces ref diffusers_helper/models/hunyuan_video_packed.py:89-112 (attn_varlen_func: the SDPA
// branch :97 is the parity target; the sageattention / flash / xformers dispatch is gone).
//
// Structure common to both sweeps in this file (one workgroup = 4 waves = 256 query rows of one head, one wave per SIMD,
// each wave owning 64 query rows as two 32-row blocks):
//   * Q stays on chip for the whole KV sweep (second generation: VGPR fragments; third: pre-scaled, in AGPRs).
//   * K and V tiles of 64 keys come by LDS-DMA (global_load_lds) into rings of 16 KiB slots.
//   * S^T = K.Q^T (operands swapped): the 32x32 accumulator then has the QUERY on the lane and 16 keys in
//     registers, so the online softmax is lane-local (one cross-half exchange when a row max is needed).
//   * O^T = V^T.P^T: the S^T accumulators, converted to bf16, ARE the B operand (no LDS round trip); V^T fragments
//     come from LDS through ds_read_b64_tr_b16 (hardware transpose), with the k index permuted to match the
//     accumulator's register->key map.
//   * K image: [key][256 B] with the 16-B chunk index XOR (key & 15)  -> conflict-free ds_read_b128.
//     V image: [d/32][key][64 B]                                      -> conflict-free tr reads.
//   * Tails: rows are clamped on load, scores of keys past the segment end are masked to -inf in the
//     last tile only, query rows past the end are not stored.  Nothing is padded in memory.
//   * The last items of a launch are cut into KV ranges ("parts": fp32 partials in a workspace, merged by
//     attn_combine_kernel), and the parts are not bound to blocks: a part block takes the next part from a counter in the
//     workspace.  The hardware deals blocks to the 8 XCDs round-robin and the dies do not run at one speed (per-workgroup
//     timeline, tools/attn_timeline.py: the same sweep takes 446 us on one XCD and 481 us on another of the same chip), so
//     with one part per block the launch ended when the slowest die had done its share and the fastest had idled for 6 %
//     of the launch; now the dies that finish their whole sweeps first take most of the parts.
// attn_w64_kernel (second generation, variant 1) and attn_v3_kernel (third, default) are described at their definitions.
#include <atomic>
#include <mutex>
#include <type_traits>
#include "fp_common.h"

namespace {

constexpr int KVBLK = 64;     // keys per tile
constexpr int HD = 128;       // head dim
constexpr int KBUF = KVBLK * HD * 2;   // 16 KiB per K (or V) buffer
constexpr int WS_FLOATS = 256 * HD + 2 * 256;   // one partial sweep: O[256][128], m[256], l[256] (fp32)
constexpr int WS_HEAD_BYTES = 256;              // the workspace starts with the part counter (int), then the partials
constexpr int PART_BLOCKS_PER_PART = 3;

__device__ __forceinline__ unsigned k_lds_off(int key, int chunk) {      // chunk = 16-B piece of the 256-B row
    return (unsigned)(key * 256 + ((chunk ^ (key & 15)) << 4));
}
__device__ __forceinline__ unsigned v_lds_off(int key, int chunk) {
    return (unsigned)((chunk >> 2) * (KVBLK * 64) + key * 64 + ((chunk & 3) << 4));
}

// value of x in lane (lane ^ 32), without LDS: v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    unsigned t = threadIdx.x;                    // opaque: the half bit is derived where it is used, not carried as a
    asm volatile("" : "+v"(t));                  // register (or a 64-bit mask) through the tile loops
    return __uint_as_float((t & 32) ? r[0] : r[1]);
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
// the same with the instruction's immediate offset, which the hardware adds to BOTH addresses
template <int OFF> __device__ __forceinline__ void dma16_at(const void* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, OFF, 0);
}
template <int OFF> __device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff,
                                                            unsigned char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, OFF, 0);
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


// Work of this block.  Blocks [0, nwg_full): one whole sweep each, XCD-contiguous.  Blocks beyond: "part blocks" - each
// takes the next part (item, KV range) from the counter at the start of the workspace, or leaves if none is left (more
// part blocks are launched than there are parts, so that a die that is ahead can take more than its round-robin share).
// Returns false if the block has nothing to do.  `smem` is used for the hand-over (before the rings are).
__device__ __forceinline__ bool pick_work(int& w, int& part, int& nparts, int nwg_full, int split, int n_parts_total,
                                          int* __restrict__ part_counter, unsigned char* smem) {
    part = 0;
    nparts = 1;
    if ((int)blockIdx.x < nwg_full) {
        w = xcd_remap(blockIdx.x, nwg_full);
        return true;
    }
    int p;
    if (part_counter) {
        if (threadIdx.x == 0)
            *reinterpret_cast<volatile int*>(smem) = __hip_atomic_fetch_add(part_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        p = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
        __syncthreads();                        // (every wave has read it before any wave's LDS-DMA lands there)
    } else {
        p = blockIdx.x - nwg_full;              // (static: the fp8 variant's launcher and callers without a counter)
    }
    if (p >= n_parts_total) return false;
    w = nwg_full + p / split;
    part = p % split;
    nparts = split;
    return true;
}

template <int THR_LOG2>
__global__ __launch_bounds__(256, 1) void attn_w64_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, bf16* __restrict__ O,
                                                          int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                                                          const int* __restrict__ cu_seqlens, int nseg, int total,
                                                          int heads, int nqb, float scale_log2e, int nwg_full,
                                                          int split, float* __restrict__ ws, int n_parts_total,
                                                          int* __restrict__ part_counter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // K ring [3][16 KiB] | V ring [3][16 KiB]
    constexpr int VRING = 3 * KBUF;
    constexpr int QBLK = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    asm volatile("" ::: "a0", "a255");          // the kernel descriptor must count the AGPRs the asm below names

    int w, part, nparts;
    if (!pick_work(w, part, nparts, nwg_full, split, n_parts_total, part_counter, smem)) return;
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

// ======================================================================================================
// Third-generation sweep (default).  The second generation's structure - one wave per SIMD, 64 query rows per wave,
// O^T in a[0:127], K / V by LDS-DMA into rings of 16 KiB slots, P.V one tile behind the softmax - with the VECTOR
// stream, the longer of the two, cut from ~6 to ~2.8 instructions per score (measured with builds that leave one stream
// out, tools/attn_skip_probe.sh: every vector instruction of the tile loop costs its full issue time - the matrix pipe
// is busy 2048 of the loop's cycles and the vector issue port nearly as long - and on a busy chip the clock falls as
// the loop gets denser, so the lever is fewer instructions, not a tighter interleave):
//   * Q is multiplied by softmax_scale * log2(e) once, when it is loaded (fp32 multiply, one more bf16 rounding of q),
//     and lives in AGPRs a[128:191] as the MFMAs' B operand;
//   * S^T accumulates straight into VGPRs, and the first MFMA of each chain takes C = -m (a 16-register block per
//     row block that holds the row's running max, negated): the accumulator IS s * scale * log2e - m, so a score costs
//     one v_exp_f32, one v_add_f32 (row sum) and half a v_cvt_pk_bf16_f32 - no AGPR read, no fma;
//   * no row max in the common case: P = exp2(s - m) is computed with the running max as it stands, and the growth test
//     rides on the row sums the softmax needs anyway: only when a tile's row sum exceeds 2^THR - some P did, the row's
//     max has grown by more than THR - is the tile redone on a slow path: true row max, m += growth, O and l rescaled,
//     the tile's and the next tile's accumulators shifted, P recomputed.  The first tile always takes it (its limit is
//     -1; m starts at 0);
//   * a lone wave per SIMD sends one instruction of any kind every ~4 cycles, and the vector ones are what the tile's
//     time follows: the tile loop is split - steps whose K(t+3) / V(t+2) tiles lie wholly inside the segment load
//     without the per-row clamp, by the BUFFER form of the LDS-DMA (lane term = loop-invariant offset register, tile term
//     = scalar offset operand, piece = immediate, four pieces per m0 value): no vector instruction per piece; the clamped
//     global form remains for the prologue, the last steps and short segments (389 -> 340 instructions per tile);
//   * every LDS fragment read is inline asm with a counted lgkmcnt in front of its first MFMA, and the first three V^T
//     fragments of the NEXT tile are requested behind this tile's last MFMAs (V runs two tiles ahead in a 4-slot ring,
//     so that tile's V is known to have landed one barrier earlier): no LDS latency is exposed behind the barrier;
//   * exp2 items are dealt one per MFMA gap, their add / pack one gap later (no wait state behind a transcendental).
//   iteration t:   MFMA stream   slots  0..31  O^T += V(t-1)^T . P(t-1)^T
//                                slots 32..63  S^T(t+1) = K(t+1) . Q'^T - m      (into the VGPR set S(t) is not in)
//                  other         slots  0..7   the 8 LDS-DMA pieces of K(t+3), V(t+2)
//                                slots  8..63  64 exp2, 64 adds, 32 packs (exp h behind MFMA 8 + 7h/8: the
//                                              P^T fragment an item completes was consumed by slot 8 (h/16) + 7)
//                                even slots    one fragment read (f + 3 while f is multiplied)
// Timing-only builds (tools/attn_skip_probe.sh): -DFP_V3_SKIP=<mask> leaves one stream of the tile loop out (results are
// then wrong): 1 exp2/pack/sum items, 2 LDS fragment reads, 4 LDS-DMA issues, 8 the per-tile barrier, 16 P.V MFMAs,
// 32 Q.K MFMAs, 128 never take the slow path after the first tile (for builds whose scores are garbage), 256 every V^T
// fragment by ONE ds_read_b128 (what a V delivered pre-transposed would cost), 512 the barrier in every other tile only
// (what 128-key iterations would save of it), 1024 no row-sum adds (the most that row sums through an MFMA against a
// ones block could save, before that MFMA's own cost).  The shipped library is built without it.
#ifndef FP_V3_SKIP
#define FP_V3_SKIP 0
#endif
// -DFP_V3_DIAG (probe builds only): shader-clock and 100 MHz real-time stamps around the whole tile loop of the first
// wave of workgroup 8 -> fp_attn_dbg = {cycles, real-time ticks, tiles, slow-path count}; read with fp_attn_dbg_read.
// In the shipped build no stamp executes.  The stamps go straight to memory: nothing of them is carried in registers
// across the loop (this kernel has none to spare).
#ifdef FP_V3_DIAG
__device__ unsigned long long fp_attn_dbg[8];
// -DFP_V3_DIAG=3: per-workgroup timeline (100 MHz real-time stamps): entry, tile loop start, tile loop end, after the
// output stores have completed; and where it ran (HW_ID | XCC_ID << 32).  Read with fp_attn_tl_read.
constexpr int FP_TL_MAX = 4096;
__device__ unsigned long long fp_attn_tl[FP_TL_MAX * 5];
#define FP_TL_STAMP(slot)                                                                                    \
    do {                                                                                                      \
        if (FP_V3_DIAG >= 3 && tid == 0 && blockIdx.x < FP_TL_MAX)                                            \
            fp_attn_tl[blockIdx.x * 5 + (slot)] = __builtin_amdgcn_s_memrealtime();                          \
    } while (0)
#else
#define FP_TL_STAMP(slot) do {} while (0)
#endif
// s = A . Q'[QIDX] + c   (first MFMA of a chain; D and C are different VGPR blocks)
template <int QIDX, int LGKM> __device__ __forceinline__ void mfma_s_first(f32x16& s, const bf16x8& a, const f32x16& c) {
    asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %4"
                 : "=&v"(s) : "v"(a), "n"(QIDX), "n"(QIDX + 3), "v"(c), "n"(LGKM));
}
template <int QIDX, int LGKM> __device__ __forceinline__ void mfma_s_next(f32x16& s, const bf16x8& a) {
    asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0"
                 : "+v"(s) : "v"(a), "n"(QIDX), "n"(QIDX + 3), "n"(LGKM));
}
template <int QIDX> __device__ __forceinline__ void mfma_s_first_nw(f32x16& s, const bf16x8& a, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %4" : "=&v"(s) : "v"(a), "n"(QIDX), "n"(QIDX + 3), "v"(c));
}
template <int QIDX> __device__ __forceinline__ void mfma_s_next_nw(f32x16& s, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(s) : "v"(a), "n"(QIDX), "n"(QIDX + 3));
}
constexpr int v3_exp_lo(int m) { return m < 8 ? 0 : (8 * (m - 8) + 6) / 7; }       // exps h with 8 + 7h/8 == m:
constexpr int v3_exp_hi(int m) { return m < 7 ? 0 : (8 * (m - 7) + 6) / 7; }       // [lo, hi)
// LDS-DMA instruction k (0..7) of a tile goes behind MFMA slot k: the gaps before the exp2 items start
#ifndef FP_V3_DMA_SLOTS
#define FP_V3_DMA_SLOTS 0, 1, 2, 3, 4, 5, 6, 7
#endif
constexpr int v3_dma_of_slot(int m) {
    constexpr int slots[8] = {FP_V3_DMA_SLOTS};
    for (int k = 0; k < 8; ++k)
        if (slots[k] == m) return k;
    return -1;
}
// LDS read instructions still in flight behind fragment f when its first MFMA issues: fragments f+1 and f+2
// (V^T fragments 0..15 take two reads, K fragments 16..31 one, 32..34 are the next tile's first V^T fragments)
constexpr int v3_reads_of(int f) { return (FP_V3_SKIP & 256) ? 1 : (f < 16 ? 2 : (f < 32 ? 1 : 2)); }
constexpr int v3_lgkm(int f) { return v3_reads_of(f + 1) + v3_reads_of(f + 2); }
// One wait per TWO fragments: the first MFMA of an even fragment f waits for f and f + 1 (requested four MFMAs ago),
// leaving only f + 2 in flight; the odd fragment's MFMAs then need no wait
constexpr int v3_lgkm2(int f) { return v3_reads_of(f + 2); }

template <int THR_LOG2>
__global__ __launch_bounds__(256, 1) void attn_v3_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                         const bf16* __restrict__ V, bf16* __restrict__ O,
                                                         int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                                                         const int* __restrict__ cu_seqlens, int nseg, int total,
                                                         int heads, int nqb, float scale_log2e, int nwg_full,
                                                         int split, float* __restrict__ ws, int n_parts_total,
                                                         int* __restrict__ part_counter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // K ring [3][16 KiB] | V ring [4][16 KiB]
    constexpr int VRING = 3 * KBUF;
    constexpr int QBLK = 256;
    constexpr int QA = 128;                     // Q' fragments: a[QA + (rb * 8 + ks) * 4 ..]
    const float rs_limit = __builtin_ldexpf(1.f, THR_LOG2);               // a tile's row sums stay below 2^THR
    const int tid = threadIdx.x, lane = tid & 63;
    FP_TL_STAMP(0);
#if defined(FP_V3_DIAG) && FP_V3_DIAG >= 3
    if (tid == 0 && blockIdx.x < FP_TL_MAX)
        fp_attn_tl[blockIdx.x * 5 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
                                         ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    asm volatile("" ::: "a0", "a255");          // the kernel descriptor must count the AGPRs the asm below names

    int w, part, nparts;
    if (!pick_work(w, part, nparts, nwg_full, split, n_parts_total, part_counter, smem)) {
#if defined(FP_V3_DIAG) && FP_V3_DIAG >= 3
        if (tid == 0 && blockIdx.x < FP_TL_MAX) fp_attn_tl[blockIdx.x * 5 + 3] = 0;      // (no work: not on the timeline)
#endif
        return;
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

    // ---- Q' = bf16(q * scale * log2e) -> AGPRs
    static_for<2>([&](auto RB) __attribute__((always_inline)) {
        const int qrow = min(q0 + wave * 64 + RB * 32 + r, seg_hi - 1);
        const bf16* qp = Q + (int64_t)qrow * q_rs + head * HD + hh * 8;
        static_for<8>([&](auto KS_) __attribute__((always_inline)) {
            const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qp + KS_ * 16);
            static_for<4>([&](auto I_) __attribute__((always_inline)) {
                const unsigned pw = cvt_pk_bf16((float)qv[2 * I_] * scale_log2e, (float)qv[2 * I_ + 1] * scale_log2e);
                agpr_write<QA + (RB * 8 + KS_) * 4 + I_>(__uint_as_float(pw));
            });
        });
    });

    // ---- LDS-DMA staging.  K: instruction i of this wave = keys wave*16 + 4 i + (0..3), lane = (key, 16-B position);
    //      the lane fetches the chunk that belongs at its position under the image's XOR.  V: instruction i = d-block
    //      `wave`, keys 16 i + (0..15), lane = (key, 16-B piece of the 64-B row).
    // Source offset of piece i = min(lane row + tile + i * 4 rows, last row) + lane column.  The tile term is scalar; the
    // lane terms are re-declared opaque at each use, else LICM keeps "row + i * pitch" and the XOR columns in a dozen
    // registers this kernel does not have (it uses all 256 VGPRs + 256 AGPRs).
    const unsigned k_step = (unsigned)(KVBLK * k_rs * 2), v_step = (unsigned)(KVBLK * v_rs * 2);
    const unsigned k_rowmax = (unsigned)((seg_hi - 1) * k_rs * 2), v_rowmax = (unsigned)((seg_hi - 1) * v_rs * 2);
    const unsigned k_i = (unsigned)(4 * k_rs * 2), v_i = (unsigned)(16 * v_rs * 2);
    const unsigned k_row0 = (unsigned)((wave * 16 + (lane >> 4)) * k_rs * 2);    // lane term of the K row
    const unsigned v_row0 = (unsigned)((lane >> 2) * v_rs * 2);                  // lane term of the V row
    unsigned k_tile = (unsigned)((seg_lo + t0 * KVBLK) * k_rs * 2);              // next K tile to load (scalar)
    unsigned v_tile = (unsigned)((seg_lo + t0 * KVBLK) * v_rs * 2);              // next V tile to load (scalar)
    const unsigned v_col = (unsigned)((head * HD + wave * 32 + (lane & 3) * 8) * 2);
    // K column of piece i: the chunk that belongs at this lane's LDS position under the image's XOR,
    // ((lane & 15) ^ ((4 i + (lane >> 4)) & 15)) * 16 B = k_col0 ^ (i << 6)
    const unsigned k_col0 = (unsigned)(head * HD * 2) + (((lane & 15) ^ (lane >> 4)) << 4);
    // INNER = the tile lies wholly inside the segment (every step of the main loop but the last few): no row needs the
    // clamp, and the lane term (row + XORed column) of a piece is one loop-invariant register: one add per piece in place
    // of add / min / xor-add (a single wave pays ~4 issue cycles for every instruction it sends, scalar ones included)
    // The four pieces of a tile go out on ONE m0 value (the slot's base): piece i reaches its LDS rows through the
    // instruction's immediate offset (i KiB), which the hardware adds to the global address as well - taken off the
    // lane term again.  (With the tile term in a scalar base address instead of the per-piece add, the compiler widens
    // the lane terms into 64-bit register pairs: sixteen registers this kernel does not have.)
    unsigned k_lane[4], v_lanes[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k_lane[i] = k_row0 + (k_col0 ^ (unsigned)(i << 6)) + i * k_i - i * 1024;     // k_i, v_i >= 1 KiB: no wrap
        v_lanes[i] = v_row0 + v_col + i * v_i - i * 1024;
    }
    // Buffer form of the LDS-DMA: address = descriptor base + lane term (register, loop-invariant) + tile term (SCALAR
    // offset operand) + immediate - no vector instruction per piece at all.
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, -1, 0x00020000);
    auto issue_k_inner = [&](auto I, int slot, unsigned tile) __attribute__((always_inline)) {
        dma16_buf<decltype(I)::value * 1024>(k_rsrc, k_lane[I], tile, smem + slot + wave * 16 * 256);
    };
    auto issue_v_inner = [&](auto I, int slot, unsigned tile) __attribute__((always_inline)) {
        dma16_buf<decltype(I)::value * 1024>(v_rsrc, v_lanes[I], tile, smem + VRING + slot + wave * (KVBLK * 64));
    };
    auto issue_k = [&](int i, int slot, auto INNER, unsigned tile) __attribute__((always_inline)) {
        unsigned off;
        {
            // (row and column terms derived again from the lane number, behind an opaque copy: kept in registers across
            //  the main loop they are three values too many and the compiler parks them in accumulator registers)
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            const unsigned row = (unsigned)((wave * 16 + (ln >> 4)) * k_rs * 2);
            const unsigned colx = (unsigned)(head * HD * 2) + ((((ln & 15) ^ (ln >> 4)) << 4) ^ (unsigned)(i << 6));
            off = min(row + (tile + i * k_i), k_rowmax) + colx;
        }
        dma16(reinterpret_cast<const unsigned char*>(K) + off, smem + slot + (wave * 16 + i * 4) * 256);
    };
    auto issue_v = [&](int i, int slot, auto INNER, unsigned tile) __attribute__((always_inline)) {
        unsigned off;
        {
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            const unsigned row = (unsigned)((ln >> 2) * v_rs * 2);
            off = min(row + (tile + i * v_i), v_rowmax) + (unsigned)((head * HD + wave * 32 + (ln & 3) * 8) * 2);
        }
        dma16(reinterpret_cast<const unsigned char*>(V) + off, smem + VRING + slot + wave * (KVBLK * 64) + i * 1024);
    };

    // ---- fragment read addresses
    const int i16 = lane & 15;
    const unsigned v_lane_off = (unsigned)(VRING + (4 * hh + (i16 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2);
    // K fragment (kb, ks) of lane (r, hh): row kb*32 + r, chunk (2 ks + hh) ^ (r & 15) = (2 ks) ^ (hh ^ (r & 15)); with the
    // lane's XOR term folded into the base, the chunk of k-step ks is base ^ (ks << 5)
    const unsigned k_lane_base = (unsigned)(r * 256) + ((unsigned)(hh ^ (r & 15)) << 4);

    // ---- state
    f32x16 S[2][2][2];                           // [tile parity][rb][kb], key of element j = kb*32 + (j&3) + 8*(j>>2) + 4*hh
    f32x16 negm[2];                              // -m_run[rb] in all 16 registers: the C operand of a chain's first MFMA
    bf16x8 pf[2][4];                             // P(t-1)/P(t): [rb][k-step]
    bf16x8 fr[4];                                // fragment ring (three fragments = 6 MFMAs ahead), carried across tiles
    float lim_cur = -1.f;                        // the growth test's limit (scalar)
    float l_run[2] = {0.f, 0.f};                 // the running max of a row block is -negm[rb][0]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int j = 0; j < 16; ++j) negm[rb][j] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[rb][s4][e] = (bf16)0.f;
    }
    asm volatile(".set fp_i, 0\n.rept 128\nv_accvgpr_write_b32 a[fp_i], 0\n.set fp_i, fp_i + 1\n.endr" ::: "memory");

    auto fence = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    // all fragment reads are asm: invisible to the compiler's lgkmcnt bookkeeping, their consumers wait explicitly
    // (the two key blocks of a k-step are read one after the other, kb = 0 first, and share the address register)
    unsigned k_addr = 0;
    auto rd_k = [&](bf16x8& f, unsigned& kbase, auto KB, auto KS_) __attribute__((always_inline)) {
        if constexpr (KB == 0) {
            asm volatile("" : "+v"(kbase));      // opaque: else the eight XORed addresses are kept in registers
            k_addr = kbase ^ (unsigned)(KS_ << 5);
        }
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(k_addr), "n"(KB * 8192));
    };
    auto rd_v = [&](bf16x8& f, unsigned vbase, auto D, auto S4) __attribute__((always_inline)) {
        union {
            u32x2 h[2];
            bf16x8 v;
        } a;
        if constexpr (FP_V3_SKIP & 256) {      // (timing only: a conflict-free b128 pattern - the K image's - inside the V slot;
            //  needs the registers that bit 4 frees: build 4 + 256 and compare with 4)
            const unsigned a = (vbase - v_lane_off) + (k_lane_base ^ (unsigned)((S4 * 2 + (D >> 1)) << 5));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"((D & 1) * 8192));
            return;
        }
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a.h[0]) : "v"(vbase), "n"(D * (KVBLK * 64) + (16 * S4) * 64));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a.h[1]) : "v"(vbase), "n"(D * (KVBLK * 64) + (16 * S4 + 8) * 64));
        f = a.v;
    };

    // ---- prologue: K(0), K(1), V(0), K(2), V(1).  Tile j of K lives in ring slot j % 3, of V in j % 4.
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(i, 0, ic<0>{}, k_tile);
    k_tile += k_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(i, KBUF, ic<0>{}, k_tile);
    k_tile += k_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_v(i, 0, ic<0>{}, v_tile);
    v_tile += v_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(i, 2 * KBUF, ic<0>{}, k_tile);
    k_tile += k_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_v(i, KBUF, ic<0>{}, v_tile);
    v_tile += v_step;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // K(0), K(1), V(0) landed
    __builtin_amdgcn_s_barrier();
    {   // S^T(0) = K(0) . Q'^T (m = 0) -> S[0]
        unsigned kbase = k_lane_base;
        static_for<8>([&](auto ks) __attribute__((always_inline)) {
            static_for<2>([&](auto kb) __attribute__((always_inline)) {
                bf16x8 f;
                rd_k(f, kbase, kb, ks);
                static_for<2>([&](auto rb) __attribute__((always_inline)) {
                    if constexpr (ks == 0) mfma_s_first<QA + (rb * 8 + ks) * 4, 0>(S[0][rb][kb], f, negm[rb]);
                    else mfma_s_next<QA + (rb * 8 + ks) * 4, 0>(S[0][rb][kb], f);
                });
            });
        });
    }
    // the first tile's P.V multiplies P = 0 with V(0) (finite data; the product must not be NaN): its first fragments
    rd_v(fr[0], v_lane_off, ic<0>{}, ic<0>{});
    rd_v(fr[1], v_lane_off, ic<1>{}, ic<0>{});
    rd_v(fr[2], v_lane_off, ic<2>{}, ic<0>{});

    // ring positions as BYTE offsets (no shift per use)
    int kslot_rd = KBUF, kslot_wr = 0;           // K(t+1) is read, K(t+3) written: for t = 0
    int vslot_rd = 0, vslot_nx = 0, vslot_wr = 2 * KBUF;   // V(t-1) (V(0) for t = 0) is read, V(t) prefetched, V(t+2) written
    // One tile step.  PAR = t & 1: S(t) sits in S[PAR], S(t+1) is accumulated into S[1 - PAR].
    auto step = [&](int t, auto PAR, auto INNER) __attribute__((always_inline)) {
        constexpr int CUR = PAR, NXT = 1 - PAR;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // all but this wave's last 8 pieces: K(t+1), V(t) are in
        if constexpr (!(FP_V3_SKIP & 8) && !((FP_V3_SKIP & 512) && PAR == 1)) __builtin_amdgcn_s_barrier();
        fence();
        unsigned kbase = (unsigned)kslot_rd + k_lane_base;
        const unsigned vbase = (unsigned)vslot_rd + v_lane_off;
        // the tile terms of this step's DMA sources, opaque: else loop strength reduction turns the eight pieces of both
        // step parities into sixteen per-lane induction registers
        unsigned kt = k_tile, vt = v_tile;
        asm volatile("" : "+s"(kt), "+s"(vt));
        float ev[4], rs[2] = {0.f, 0.f};     // exp ring: at most four values wait for their add / pack
        // row-sum add of exp h and, with the second value of a pair, the pack into its P^T dword
        auto finish_exp = [&](auto H_) __attribute__((always_inline)) {
            constexpr int h = H_, i = h >> 1, s4 = i >> 3, rb = (i >> 2) & 1, dw = i & 3;
            if constexpr (!(FP_V3_SKIP & 1024)) {
                rs[rb] += ev[h & 3];
                asm volatile("" : "+v"(rs[rb]));
            }
            if constexpr (h & 1) {
                union {
                    unsigned u[4];
                    bf16x8 v;
                } pk;
                pk.v = pf[rb][s4];
                unsigned pw = cvt_pk_bf16(ev[(h - 1) & 3], ev[h & 3]);
                asm volatile("" : "+v"(pw));
                pk.u[dw] = pw;
                pf[rb][s4] = pk.v;
            }
        };
        if (!INNER && (t0 + t == nt_all - 1) && (nkv & (KVBLK - 1))) {          // ragged last tile: keys past the end score -inf
            // the lane's first invalid key offset, opaque and defined in here: hoisted out of the loop, the 32 compare
            // masks would sit in 64 SGPRs for the whole sweep
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            int lim = nkv - (t0 + t) * KVBLK - 4 * (int)(ln >> 5);
            asm volatile("s_nop 15\ns_nop 7" : "+v"(lim)::"memory");  // (and the last MFMAs of S(t) have retired)
#pragma unroll
            for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (kb2 * 32 + (j & 3) + 8 * (j >> 2) >= lim) S[CUR][rb2][kb2][j] = -INFINITY;
        }
        fence();
        static_for<64>([&](auto M_) __attribute__((always_inline)) {
            constexpr int m = M_;
            constexpr int f = m >> 1;                                 // fragment of this slot (0..15 V^T, 16..31 K)
            // ---------------- the MFMA of this slot
            if constexpr (m < 32) {
                constexpr int s4 = m >> 3, d = (m >> 1) & 3, rb = m & 1;
                if constexpr (FP_V3_SKIP & 16) asm volatile("" ::"v"(fr[f & 3]), "v"(pf[rb][s4]));
                else if constexpr (rb == 0 && (f & 1) == 0) mfma32_after<(d * 2 + rb) * 16, v3_lgkm2(f)>(fr[f & 3], pf[rb][s4]);
                else mfma32<(d * 2 + rb) * 16, false>(fr[f & 3], pf[rb][s4]);
            } else {
                constexpr int q = m - 32, ks = q >> 2, kb = (q >> 1) & 1, rb = q & 1;
                if constexpr (FP_V3_SKIP & 32) asm volatile("" : "+v"(S[NXT][rb][kb]) : "v"(fr[f & 3]));
                else if constexpr (rb == 0 && (f & 1) == 0 && ks == 0) mfma_s_first<QA + (rb * 8 + ks) * 4, v3_lgkm2(f)>(S[NXT][rb][kb], fr[f & 3], negm[rb]);
                else if constexpr (rb == 0 && (f & 1) == 0) mfma_s_next<QA + (rb * 8 + ks) * 4, v3_lgkm2(f)>(S[NXT][rb][kb], fr[f & 3]);
                else if constexpr (ks == 0) mfma_s_first_nw<QA + (rb * 8 + ks) * 4>(S[NXT][rb][kb], fr[f & 3], negm[rb]);
                else mfma_s_next_nw<QA + (rb * 8 + ks) * 4>(S[NXT][rb][kb], fr[f & 3]);
            }
            fence();
            // ---------------- LDS reads: fragment f+3 while fragment f is multiplied (one fragment per two slots);
            // fragments 32..34 are the first three V^T fragments of the next tile's P.V
            if constexpr ((m & 1) == 0 && !(FP_V3_SKIP & 2)) {
                constexpr int f2 = f + 3;
                if constexpr (f2 < 16) rd_v(fr[f2 & 3], vbase, ic<f2 & 3>{}, ic<(f2 >> 2)>{});
                else if constexpr (f2 < 32) rd_k(fr[f2 & 3], kbase, ic<(f2 - 16) & 1>{}, ic<((f2 - 16) >> 1)>{});
                else rd_v(fr[f2 & 3], (unsigned)vslot_nx + v_lane_off, ic<(f2 - 32) & 3>{}, ic<0>{});
            }
            // ---------------- LDS-DMA: K(t+3) and V(t+2)
            if constexpr (v3_dma_of_slot(m) >= 0 && !(FP_V3_SKIP & 4)) {
                constexpr int i = v3_dma_of_slot(m);
                if constexpr (INNER) {
                    if constexpr (i < 4) issue_k_inner(ic<i>{}, kslot_wr, kt);
                    else issue_v_inner(ic<i - 4>{}, vslot_wr, vt);
                } else {
                    if constexpr (i < 4) issue_k(i, kslot_wr, INNER, kt);
                    else issue_v(i - 4, vslot_wr, INNER, vt);
                }
            }
            // ---------------- P(t) = exp2(S(t)): exp h of the tile (two values per P^T dword, in the order the k-steps use
            // them) is issued in gap 8 + 7h/8; its row-sum add and the pack of its pair follow one gap LATER, so that no
            // instruction waits for a transcendental's result (a dependent neighbour costs an s_nop)
            static_for<(FP_V3_SKIP & 1) ? 0 : v3_exp_hi(m) - v3_exp_lo(m)>([&](auto HH) __attribute__((always_inline)) {
                constexpr int h = v3_exp_lo(m) + HH, i = h >> 1, s4 = i >> 3, rb = (i >> 2) & 1, dw = i & 3;
                constexpr int kb = s4 >> 1, j = (s4 & 1) * 8 + 2 * dw + (h & 1);
                ev[h & 3] = __builtin_amdgcn_exp2f(S[CUR][rb][kb][j]);
                asm volatile("" : "+v"(ev[h & 3]));                   // pinned to this gap (volatile asms keep their order)
            });
            if constexpr (m >= 1)
                static_for<(FP_V3_SKIP & 1) ? 0 : v3_exp_hi(m - 1) - v3_exp_lo(m - 1)>([&](auto HH) __attribute__((always_inline)) {
                    finish_exp(ic<v3_exp_lo(m - 1) + HH>{});
                });
            fence();
        });
        if constexpr (!(FP_V3_SKIP & 1)) finish_exp(ic<63>{});
        // ---------------- did a row's max grow by more than THR?  Then some P = exp2(s - m) of that row exceeds 2^THR and
        // so does the tile's row sum (an overflow to inf or a NaN fails the test too): redo the tile on the slow path
        // (one max, one compare, one branch on vcc; a NaN sum needs no redo - the row is NaN either way)
        // The first tile always goes this way: its limit is -1 (row sums are >= 0), set to 2^THR in there.
        bool slow = __builtin_amdgcn_ballot_w64(!(fmaxf(rs[0], rs[1]) <= lim_cur)) != 0;
        if constexpr (FP_V3_SKIP & 128) slow = t == 0;
        if (slow) {
            const bool first = t == 0;
            lim_cur = rs_limit;
            asm volatile("s_nop 15\ns_nop 15\ns_nop 7" ::: "memory"); // every MFMA of this iteration has retired
            float alpha[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float a0 = S[CUR][rb][0][0], a1 = S[CUR][rb][1][0];
#pragma unroll
                for (int j = 1; j < 16; ++j) {
                    a0 = fmaxf(a0, S[CUR][rb][0][j]);
                    a1 = fmaxf(a1, S[CUR][rb][1][j]);
                }
                float mx = fmaxf(a0, a1);
                mx = fmaxf(mx, other_half(mx));
                // the accumulators hold s - m: the row's max has grown by mx (if positive); rows that did not grow keep m
                const float delta = first ? mx : fmaxf(mx, 0.f);
                alpha[rb] = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
                l_run[rb] *= alpha[rb];
                const float nm = negm[rb][0] - delta;
#pragma unroll
                for (int j = 0; j < 16; ++j) negm[rb][j] = nm;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        S[CUR][rb][kb][j] -= delta;
                        S[NXT][rb][kb][j] -= delta;                   // S(t+1) was accumulated onto the old -m
                    }
            }
            rs[0] = rs[1] = 0.f;
            static_for<32>([&](auto I_) __attribute__((always_inline)) {
                constexpr int i = I_, s4 = i >> 3, rb = (i >> 2) & 1, dw = i & 3, kb = s4 >> 1, j = (s4 & 1) * 8 + 2 * dw;
                const float e0 = __builtin_amdgcn_exp2f(S[CUR][rb][kb][j]);
                const float e1 = __builtin_amdgcn_exp2f(S[CUR][rb][kb][j + 1]);
                rs[rb] += e0 + e1;
                union {
                    unsigned u[4];
                    bf16x8 v;
                } pk;
                pk.v = pf[rb][s4];
                pk.u[dw] = cvt_pk_bf16(e0, e1);
                pf[rb][s4] = pk.v;
            });
            if (!first) {                                             // O *= alpha (V(t-1) has been added)
                static_for<128>([&](auto I_) __attribute__((always_inline)) {
                    constexpr int i = I_, rb = (i >> 4) & 1;
                    agpr_write<i>(agpr_read<i>() * alpha[rb]);
                });
            }
#if defined(FP_V3_DIAG) && FP_V3_DIAG > 1
            if (blockIdx.x == 8 && tid == 0) fp_attn_dbg[3] += 1;      // (-DFP_V3_DIAG=2: needs registers a full build lacks)
#endif
        }
        l_run[0] += rs[0];
        l_run[1] += rs[1];
        k_tile += k_step;
        v_tile += v_step;
        kslot_wr = kslot_rd;                                          // K(t+4) goes where K(t+1) was
        kslot_rd = kslot_rd == 2 * KBUF ? 0 : kslot_rd + KBUF;
        vslot_rd = vslot_nx;                                          // next: P.V reads V(t), prefetches V(t+1), loads V(t+3)
        vslot_nx = (vslot_nx + KBUF) & (4 * KBUF - 1);
        vslot_wr = (vslot_wr + KBUF) & (4 * KBUF - 1);
        // The compiler does not know that the asm above writes S(t+1) with a 16-pass MFMA: a register copy it places
        // behind this step (it does, on the way from the loop to the odd last step) must not come within 18 wait
        // states of the last MFMA.  The end-of-tile check above covers about half of them.
        asm volatile("s_nop 11" ::: "memory");
        // ... and the registers of S(t+1) must stay allocated until the MFMAs have written them even where the value is
        // dead (the last step: S(t+1) is never used, and the compiler handed its registers to the exp2 items while the
        // MFMAs that write them were still in flight)
        asm volatile("" ::"v"(S[NXT][0][0]), "v"(S[NXT][0][1]), "v"(S[NXT][1][0]), "v"(S[NXT][1][1]));
    };
#ifdef FP_V3_DIAG
    if (blockIdx.x == 8 && tid == 0) {
        fp_attn_dbg[4] = __builtin_amdgcn_s_memtime();
        fp_attn_dbg[5] = __builtin_amdgcn_s_memrealtime();
        fp_attn_dbg[3] = 0;
    }
#endif
    FP_TL_STAMP(1);
    {
        // step t loads K(t+3) and V(t+2) (tiles counted from t0): all of their rows are the segment's own while
        // t0 + t + 3 < nkv / 64.  Those steps run without the per-row clamp; the last few (and a short segment) with it.
        const int n_inner = max(min(nt, nkv / KVBLK - t0 - 3), 0) & ~1;
        int t = 0;
        for (; t < n_inner; t += 2) {
            step(t, ic<0>{}, ic<1>{});
            step(t + 1, ic<1>{}, ic<1>{});
        }
        for (; t + 1 < nt; t += 2) {
            step(t, ic<0>{}, ic<0>{});
            step(t + 1, ic<1>{}, ic<0>{});
        }
        if (t < nt) step(t, ic<0>{}, ic<0>{});
    }
#ifdef FP_V3_DIAG
    if (blockIdx.x == 8 && tid == 0) {
        fp_attn_dbg[0] = __builtin_amdgcn_s_memtime() - fp_attn_dbg[4];
        fp_attn_dbg[1] = __builtin_amdgcn_s_memrealtime() - fp_attn_dbg[5];
        fp_attn_dbg[2] = (unsigned long long)nt;
    }
#endif
    FP_TL_STAMP(2);
    // ---- the last tile's P.V (its V tile landed before the last barrier; the prefetched fragments are read again)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(fr[0]), "v"(fr[1]), "v"(fr[2]));    // registers of the in-flight prefetch stay allocated until here
    {
        const unsigned vbase = (unsigned)vslot_rd + v_lane_off;
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
    // lane terms of the output rows, derived again behind an opaque copy: shared with the prologue's they would be carried
    // through the tile loops, which have no register to spare
    unsigned ln_e = (unsigned)lane;
    asm volatile("" : "+v"(ln_e));
    const int r_e = (int)(ln_e & 31), hh_e = (int)(ln_e >> 5);
    // (uniform; through a lane read so that it is not shared with the prologue's per-lane row, which would stay live)
    const int qbase_e = __builtin_amdgcn_readfirstlane(q0 + wave * 64);
    if (nparts > 1) {                           // partial sweep: unnormalised O (fp32), running max and sum
        static_for<2>([&](auto rb) __attribute__((always_inline)) {
            const int row = wave * 64 + rb * 32 + r_e;
            float* op = wsp + row * HD + 4 * hh_e;
            static_for<4>([&](auto d) __attribute__((always_inline)) {
                static_for<4>([&](auto g) __attribute__((always_inline)) {
                    f32x4 v;
                    static_for<4>([&](auto j) __attribute__((always_inline)) { v[(int)j] = agpr_read<(d * 2 + rb) * 16 + 4 * g + j>(); });
                    *reinterpret_cast<f32x4*>(op + d * 32 + g * 8) = v;
                });
            });
            if (hh_e == 0) {
                wsp[256 * HD + row] = -negm[rb][0];
                wsp[256 * HD + 256 + row] = l_tot[rb];
            }
        });
#if defined(FP_V3_DIAG) && FP_V3_DIAG >= 3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FP_TL_STAMP(3);
#endif
        return;
    }
    static_for<2>([&](auto rb) __attribute__((always_inline)) {
        const float inv = 1.f / l_tot[rb];
        const int qrow_real = qbase_e + rb * 32 + r_e;
        if (qrow_real < seg_hi) {
            bf16* op = O + (int64_t)qrow_real * o_rs + head * HD + 4 * hh_e;
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
#if defined(FP_V3_DIAG) && FP_V3_DIAG >= 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FP_TL_STAMP(3);
#endif
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
    // (an atomic: the first calls may come from two threads at once - the reference runs forward() on a worker thread)
    static std::atomic<int> n{0};
    int v = n.load(std::memory_order_relaxed);
    if (v == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n.store(v, std::memory_order_relaxed);
    }
    return v;
}

}  // namespace

// Diagnostics switch (same-process A/B, the tests of the generation that is not the default).  An atomic, read once
// per launch: the reference calls forward() from a worker thread.
static std::atomic<int> g_attn_variant{2};      // 1: second-generation sweep (S^T in AGPRs) | 2: third generation (S^T - m in VGPRs); + 16: static parts
extern "C" int fp_debug_attn_variant(int v) {
    return v >= 0 ? g_attn_variant.exchange(v, std::memory_order_relaxed) : g_attn_variant.load(std::memory_order_relaxed);
}

// Library-internal (attn_fp8.hip): the merge of split tail items and the CU count behind the split decision.
__attribute__((visibility("hidden"))) int fpi_attn_cu_count() { return cu_count(); }
__attribute__((visibility("hidden"))) int fpi_attn_combine(const float* ws, void* o, int64_t o_rs, const int* cu_seqlens,
                                                           int total, int heads, int nqb, int nwg_full, int split,
                                                           int ntail, hipStream_t st) {
    attn_combine_kernel<<<dim3((unsigned)(ntail * 16)), dim3(256), 0, st>>>(ws, (bf16*)o, o_rs, cu_seqlens, total, heads,
                                                                            nqb, nwg_full, split);
    FP_CHECK_LAUNCH();
    return 0;
}

// Bytes of device workspace with which fp_attn_fwd_bf16 can split the last round's work items (any shape).
// (two partial slots per CU: a partial last round of up to all the CUs' worth of items can be cut at least in two)
extern "C" int64_t fp_attn_workspace_bytes(void) { return WS_HEAD_BYTES + 2 * (int64_t)cu_count() * WS_FLOATS * 4; }

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
    // One workgroup per CU is resident (512 registers per lane): nwg items take ceil(nwg / CUs) rounds.  With a workspace,
    // the launch's last `tail` items are each cut into `split` KV ranges (parts), swept by separate part blocks and merged:
    // a partly filled last round then takes about 1 / split of a sweep instead of a whole one, and - the parts being
    // taken from a counter, not bound to blocks - the dies that are ahead take more of them (see the head of this file).
    // tail = the items of the partial last round (r), cut as finely as the workspace's partial slots allow (at most 8
    // ways); when that round holds an eighth of the CUs' worth or less - also none at all: the dies still differ - an
    // eighth of the CUs' worth (32 items = 256 parts on an MI355X); when cutting the whole partial round would not pay
    // (too few slots for r items to be cut at all, or so few ways that the parts take a sweep's time again), that eighth
    // as well.
    const int ncu = cu_count();
    const int variant = g_attn_variant.load(std::memory_order_relaxed);      // read once per launch
    const bool static_parts = (variant & 16) != 0;     // diagnostics (A/B): round 2's split - one block per part, only a partial round <= half the CUs is split
    int nwg_full = (int)nwg, split = 1, tail = 0;
    const int64_t slots = workspace && ((uintptr_t)workspace & 15) == 0 ? (workspace_bytes - WS_HEAD_BYTES) / ((int64_t)WS_FLOATS * 4) : 0;
    if (slots > 0 && nwg > ncu && ncu >= 8 && !(static_parts && (nwg % ncu == 0 || nwg % ncu > ncu / 2))) {
        const int r = (int)(nwg % ncu), base = ncu / 8;
        tail = static_parts ? r : (r > base ? r : base);
        split = (int)(slots / tail < 8 ? slots / tail : 8);
        // (the parts of a round cut `split` ways take ceil(parts / CUs) / split of a sweep: two rounds of halves are
        // the whole sweep again - measured: 200 items cut in two are 1 % slower than the eighth's split, 144 items cut in
        // three 4.4 % faster)
        if (!static_parts && (split < 2 || ((int64_t)tail * split + ncu - 1) / ncu * 10 > 9 * split)) {
            tail = base;
            split = (int)(slots / tail < 8 ? slots / tail : 8);
        }
        if (static_parts) split = ncu / tail < 8 ? ncu / tail : 8;
        if (split >= 2 && (int64_t)tail * split <= slots)
            nwg_full = (int)nwg - tail;
        else if (workspace_required)
            return (int)hipErrorInvalidValue;       // the fixed scratch cannot hold this device's parts
        else
            split = 1, tail = 0;
    } else if (workspace_required && workspace && nwg > ncu && slots <= 0) {
        return (int)hipErrorInvalidValue;
    }
    const int n_parts = (int)(nwg - nwg_full) * split;
    // three part blocks per part: a die 8 % ahead of the slowest one takes about twice its round-robin share
    const int64_t grid = nwg_full + (int64_t)n_parts * (static_parts ? 1 : PART_BLOCKS_PER_PART);
    FP_REQUIRE(grid < (1ll << 31));
    int* counter = n_parts && !static_parts ? (int*)workspace : nullptr;
    float* partials = n_parts ? (float*)((char*)workspace + WS_HEAD_BYTES) : nullptr;
    if (counter) {
        hipError_t e = hipMemsetAsync(counter, 0, sizeof(int), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    // defer-max threshold.  Second generation: 2^8 (P <= 256 before the bf16 rounding, the O rescale runs only when a
    // row's max grows by more than 8 in the log2 domain).  Third generation: 2^60 - the threshold only has to keep
    // fp32 from overflowing (P <= 2^60, row sums and O below 2^60 * keys * |v|): bf16 and fp32 keep their relative
    // precision at any magnitude, and a stale m never makes a P smaller than the true max would
    {
        const bool gen2 = (variant & 3) == 1;
        auto kern = gen2 ? attn_w64_kernel<8> : attn_v3_kernel<60>;
        const int lds = (gen2 ? 6 : 7) * KBUF;      // K ring of 3 slots + V ring of 3 (second generation) or 4 slots
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        kern<<<dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream>>>(
            (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, q_row_stride, k_row_stride, v_row_stride,
            o_row_stride, cu_seqlens, nseg, total_tokens, heads, nqb, scale_log2e, nwg_full, split, partials, n_parts,
            counter);
    }
    FP_CHECK_LAUNCH();
    if (nwg_full < nwg)
        attn_combine_kernel<<<dim3((unsigned)((nwg - nwg_full) * 16)), dim3(256), 0, (hipStream_t)stream>>>(
            partials, (bf16*)o, o_row_stride, cu_seqlens, total_tokens, heads, nqb, nwg_full, split);
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
static __device__ __align__(256) float scratch[WS_HEAD_BYTES / 4 + (size_t)SCRATCH_PARTIALS * WS_FLOATS];
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

extern "C" int64_t fp_attn_static_scratch_bytes(void) { return WS_HEAD_BYTES + (int64_t)SCRATCH_PARTIALS * WS_FLOATS * 4; }

extern "C" int fp_attn_fwd_bf16_static(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                                       int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                                       const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens, int heads,
                                       int head_dim, float softmax_scale, void* stream) {
    float* ptr = nullptr;
    const int e = scratch_for_current_device(&ptr);
    if (e) return e;
    // every partial the kernels read is written by the same call first (attn_*_kernel stores O, m, l of each part
    // before attn_combine_kernel runs); the part counter at the array's start is cleared by the launcher (a memset node
    // on the stream: no allocation)
    return attn_launch(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, cu_seqlens, nseg, max_seqlen,
                       total_tokens, heads, head_dim, softmax_scale, ptr, fp_attn_static_scratch_bytes(), true, stream);
}

#ifdef FP_V3_DIAG
extern "C" int fp_attn_dbg_read(unsigned long long* out4) {
    return (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(fp_attn_dbg), sizeof(unsigned long long) * 4);
}
extern "C" int fp_attn_tl_read(unsigned long long* out, int n_workgroups) {
    if (n_workgroups > FP_TL_MAX) n_workgroups = FP_TL_MAX;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fp_attn_tl), sizeof(unsigned long long) * 5 * n_workgroups);
}
#endif
