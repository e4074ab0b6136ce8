// K5: C = epilogue(A . W^T + bias), bf16 in / fp32 accumulate / bf16 out  (gfx950 MFMA 32x32x16, wave64)
//
// The nn.Linear of every DiT block (ref hunyuan_video_packed.py:119-121, 133-135, 154-156, 167-169, 552,
// 570 and the diffusers FeedForward it calls at :646-647), with the elementwise work that follows each of
// them in the reference fused into the epilogue: bias, tanh-GELU (optionally only on a column range: the
// fused [k|v|q|mlp] projection of the single-stream blocks) and the gated residual `R + gate * (.)`.
//
// Both operands are K-contiguous ("NT"): A [M,K] tokens x features, W [N,K] nn.Linear layout.
// Tile 256(M) x 256(N) x 64(K), 8 waves as 2(M) x 4(N), wave tile 128 x 64 = 4 x 2 MFMA 32x32 blocks.
// The MFMA is issued with the operands swapped (D^T = W.A^T): the accumulator then has the TOKEN ROW on
// the lane and 4 consecutive output columns in consecutive registers, so the epilogue stores 8 B per lane
// contiguous in N and bias/gate are per-register constants.
// Staging: global -> registers -> LDS, one register set, LDS double buffered; tile t+1 is written right
// after the barrier and the loads of tile t+2 are issued before the MFMAs of tile t (one barrier per K tile).
// LDS image per operand: [row][128 B] with the 16-B chunk index XOR ((row >> 1) & 7): ds_read_b128
// fragment reads and the ds_write_b128 staging writes are both bank-conflict free.
#include "fp_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB per operand per buffer

__device__ __forceinline__ unsigned lds_off(int row, int chunk) {
    return (unsigned)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

__device__ __forceinline__ float gelu_tanh(float v) {
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    const float e = __expf(2.f * u);
    const float th = 1.f - 2.f / (e + 1.f);
    return 0.5f * v * (1.f + th);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(const bf16* __restrict__ A, int64_t lda,
                                                           const bf16* __restrict__ A2, int k1,
                                                           const bf16* __restrict__ W, int64_t ldw,
                                                           const bf16* __restrict__ bias, bf16* __restrict__ C,
                                                           int64_t ldc, int M, int N, int K, int gelu_from,
                                                           const bf16* __restrict__ R, int64_t ldr,
                                                           const bf16* __restrict__ gate, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* abuf = smem;                      // [2][TILE_BYTES]
    unsigned char* wbuf = smem + 2 * TILE_BYTES;     // [2][TILE_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // tile order: XCD-contiguous, grouped 8 row-tiles tall so that co-running tiles share A and W panels
    constexpr int GM = 8;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GM * ntn;
    const int group = wid / per_group, in_group = wid % per_group;
    const int first_m = group * GM;
    const int gsz = min(ntm - first_m, GM);
    const int tm = first_m + in_group % gsz, tn = in_group / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging map: 2048 16-B chunks per operand tile, four per thread: row = (tid >> 3) + 64 i, chunk = tid & 7
    const int srow = tid >> 3, sch = tid & 7;
    const bf16* aptr[4];
    const bf16* a2ptr[4];
    const bf16* wptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ar = min(m0 + srow + 64 * i, M - 1);
        const int wr = min(n0 + srow + 64 * i, N - 1);
        aptr[i] = A + (int64_t)ar * lda + sch * 8;
        a2ptr[i] = A2 ? (A2 + (int64_t)ar * lda + sch * 8 - k1) : aptr[i];
        wptr[i] = W + (int64_t)wr * ldw + sch * 8;
    }
    u32x4 areg[4], wreg[4];
    auto stage_load = [&](int kt) {
        const int k0 = kt * BK;
        const bool second = k0 >= k1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            areg[i] = *reinterpret_cast<const u32x4*>((second ? a2ptr[i] : aptr[i]) + k0);
            wreg[i] = *reinterpret_cast<const u32x4*>(wptr[i] + k0);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(abuf + buf * TILE_BYTES + lds_off(srow + 64 * i, sch)) = areg[i];
            *reinterpret_cast<u32x4*>(wbuf + buf * TILE_BYTES + lds_off(srow + 64 * i, sch)) = wreg[i];
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    const int nk = K / BK;
    stage_load(0);
    stage_write(0);
    if (nk > 1) stage_load(1);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt > 0) {
            __syncthreads();                              // everyone is done reading buf[cur^1] (tile kt-1)
        }
        if (kt + 1 < nk) {
            stage_write(cur ^ 1);                         // registers hold tile kt+1
            if (kt + 2 < nk) stage_load(kt + 2);
        }
        const unsigned char* ap = abuf + cur * TILE_BYTES;
        const unsigned char* wp = wbuf + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[4], wf[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                wf[ni] = *reinterpret_cast<const bf16x8*>(wp + lds_off(wn * 64 + ni * 32 + r, 2 * ks + hh));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(ap + lds_off(wm * 128 + mi * 32 + r, 2 * ks + hh));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
    }
    // NOTE on the loop above: tile kt+1 is written to buf[cur^1] AFTER the barrier at the top of iteration
    // kt and read in iteration kt+1 after the barrier at ITS top; the write in iteration kt can race with no
    // read because every wave passed the top barrier only after finishing tile kt-1 (the last reader of
    // buf[cur^1]).

    // ---- epilogue: lane (r,hh) holds row m = m0 + wm*128 + mi*32 + r, columns n0 + wn*64 + ni*32 + 8g + 4hh + (0..3)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 128 + mi * 32 + r;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * hh;
                if (n >= N) continue;
                float v[4];
                bf16x4 bv = {0, 0, 0, 0};
                if (bias) bv = *reinterpret_cast<const bf16x4*>(bias + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * g + j] + (float)bv[j];
                if (EPI == 1 || (EPI == 2 && n >= gelu_from)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_tanh(v[j]);
                }
                if (EPI == 3) {
                    const bf16x4 gv = *reinterpret_cast<const bf16x4*>(gate + n);
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + (int64_t)m * ldr + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)rv[j] + (float)gv[j] * v[j];
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
                *reinterpret_cast<bf16x4*>(C + (int64_t)m * ldc + n) = o;
            }
        }
    }
}

template <int EPI>
int launch(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw, const void* bias,
           void* c, int64_t ldc, int M, int N, int K, int gelu_from, const void* resid, int64_t ldr,
           const void* gate, hipStream_t st) {
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    constexpr int smem = 4 * TILE_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    gemm_bf16_kernel<EPI><<<dim3((unsigned)(ntm * ntn)), dim3(512), smem, st>>>(
        (const bf16*)a, lda, (const bf16*)a2, k1, (const bf16*)w, ldw, (const bf16*)bias, (bf16*)c, ldc, M, N, K,
        gelu_from, (const bf16*)resid, ldr, (const bf16*)gate, ntm, ntn);
    FP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int fp_gemm_bf16(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw,
                            const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue,
                            int gelu_from, const void* resid, int64_t ldr, const void* gate, void* stream) {
    FP_REQUIRE(a && w && c && M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0);
    FP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && ldw >= K && ldc >= N);
    FP_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 7) == 0);
    FP_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0);
    if (a2) {
        FP_REQUIRE(k1 > 0 && k1 < K && k1 % BK == 0 && ((uintptr_t)a2 & 15) == 0 && lda >= k1 && lda >= K - k1);
    } else {
        FP_REQUIRE(lda >= K);
        k1 = K;
    }
    FP_REQUIRE(epilogue >= 0 && epilogue <= 3);
    if (epilogue == 2) FP_REQUIRE(gelu_from >= 0 && gelu_from % 4 == 0);
    if (epilogue == 3)
        FP_REQUIRE(resid && gate && ldr % 4 == 0 && ldr >= N && ((uintptr_t)resid & 7) == 0 && ((uintptr_t)gate & 7) == 0);
    if (M == 0) return 0;
    FP_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31));
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case 0: return launch<0>(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, st);
        case 1: return launch<1>(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, st);
        case 2: return launch<2>(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, st);
        default: return launch<3>(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, st);
    }
}
