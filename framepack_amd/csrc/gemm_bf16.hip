// K5: the nn.Linear of every DiT block, bf16 (kernel: gemm_pipeline.h).
#include <atomic>
#include "gemm_w4.h"

// Diagnostics switch (A/B measurements, the tests of the kernel generation that is not the default).  An atomic: the
// reference calls forward() from a worker thread, and a launch reads the value exactly once.
static std::atomic<int> g_variant_switch{1};      // 0: 8-wave ping-pong pipeline | 1: one wave per SIMD
__attribute__((visibility("hidden"))) int fpi_gemm_variant() { return g_variant_switch.load(std::memory_order_relaxed); }      // gemm_fp8.hip follows the same switch
extern "C" int fp_debug_gemm_variant(int v) {
    return v >= 0 ? g_variant_switch.exchange(v, std::memory_order_relaxed) : g_variant_switch.load(std::memory_order_relaxed);
}

extern "C" int fp_gemm_bf16(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw,
                            const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue,
                            int gelu_from, const void* resid, int64_t ldr, const void* gate, void* stream) {
    using namespace fp_gemm;
    int64_t a2_delta;
    if (int rc = check_dense(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, epilogue, gelu_from, resid, ldr, gate, a2_delta))
        return rc;
    if (M == 0) return 0;
    const int g_variant = fpi_gemm_variant();           // read once per launch
    ConvGeom cg{};
    cg.gm = K >= 8192 ? -3 : (N >= 16384 ? 2 : 8);      // see ConvGeom::gm
    if (g_variant >> 4) cg.gm = (g_variant & 8) ? -(g_variant >> 4) : (g_variant >> 4);     // diagnostics: 1 | gm << 4 (| 8: column-major walk)
    hipStream_t st = (hipStream_t)stream;
    // the 4-wave kernel moves output (and residual) rows 16 B per lane; operands aligned only to the 8 B the
    // contract asks for go to the 8-wave kernel
    const bool wide = ((uintptr_t)c & 15) == 0 && ldc % 8 == 0 && (epilogue < 3 || (((uintptr_t)resid & 15) == 0 && ldr % 8 == 0));
    const bool w4 = (g_variant & 1) && wide;
    // Fewer 256 x 256 tiles than half the CUs (the per-rank row counts of 8-way sequence parallelism, the text
    // stream): 256 x 128 tiles double the number of busy CUs at ~0.6 of the time per tile
    const bool narrow = w4 && N > 128 &&
                        (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) <= persistent_grid() / 2;
#define FP_CASE(E)                                                                                                  \
    case E:                                                                                                         \
        if (narrow)                                                                                                 \
            return launch_w4<bf16, E, 0, bf16, 4>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from,  \
                                                  resid, ldr, gate, cg, st);                                        \
        return w4 ? launch_w4<bf16, E, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from,    \
                                                 resid, ldr, gate, cg, st)                                          \
                         : launch<bf16, E, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, \
                                              ldr, gate, cg, st)
    switch (epilogue) {
        FP_CASE(0);
        FP_CASE(1);
        FP_CASE(2);
        FP_CASE(3);
        default: FP_CASE(4);
    }
#undef FP_CASE
}

// Two nn.Linear of one shape over the two row segments of one activation buffer, in ONE launch: rows [0, m_split) of a
// multiply w0 (+ bias0, gate0), rows [m_split, M) multiply w1 (+ bias1, gate1); c, resid likewise one buffer each.  The
// double-stream blocks (ref hunyuan_video_packed.py:537-604: to_q/k/v | add_q/k/v_proj, to_out | to_add_out, ff | ff_context)
// keep the image rows and the text rows in one buffer; the text segment alone is 4 - 16 row tiles, a launch that fills a
// fraction of the chip for a whole tile time.  Same arithmetic per output element as two fp_gemm_bf16 calls (tiles never
// straddle m_split): bit-identical.  Operands that do not meet the 4-wave kernel's alignment, a segment that is empty and
// segments that would each run on the narrow tile take the two calls instead.
extern "C" int fp_gemm_bf16_rows2(const void* a, int64_t lda, const void* w0, const void* w1, int64_t ldw,
                                  const void* bias0, const void* bias1, void* c, int64_t ldc, int M, int m_split, int N,
                                  int K, int epilogue, int gelu_from, const void* resid, int64_t ldr, const void* gate0,
                                  const void* gate1, void* stream) {
    using namespace fp_gemm;
    FP_REQUIRE(m_split >= 0 && m_split <= M && w1 && (!bias0 == !bias1) && (!gate0 == !gate1));
    int64_t a2_delta;
    int k1 = 0;
    if (int rc = check_dense(a, lda, nullptr, k1, w0, ldw, bias0, c, ldc, M, N, K, epilogue, gelu_from, resid, ldr, gate0, a2_delta))
        return rc;
    FP_REQUIRE(((uintptr_t)w1 & 15) == 0 && (!bias1 || ((uintptr_t)bias1 & 7) == 0) && (!gate1 || ((uintptr_t)gate1 & 7) == 0));
    FP_REQUIRE(epilogue == 0 || epilogue == 1 || epilogue == 3);
    if (M == 0) return 0;
    const bool wide = ((uintptr_t)c & 15) == 0 && ldc % 8 == 0 && (epilogue < 3 || (((uintptr_t)resid & 15) == 0 && ldr % 8 == 0));
    const int64_t tiles = ((int64_t)(m_split + BM - 1) / BM + (M - m_split + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (!(fpi_gemm_variant() & 1) || (fpi_gemm_variant() & 2) || !wide || m_split == 0 || m_split == M ||
        tiles <= persistent_grid() / 2) {          // (variant bit 1: diagnostics - the two launches)
        const char* a1 = (const char*)a + (int64_t)m_split * lda * 2;
        char* c1 = (char*)c + (int64_t)m_split * ldc * 2;
        const char* r1 = resid ? (const char*)resid + (int64_t)m_split * ldr * 2 : nullptr;
        if (m_split > 0)
            if (int rc = fp_gemm_bf16(a, lda, nullptr, 0, w0, ldw, bias0, c, ldc, m_split, N, K, epilogue, gelu_from, resid, ldr, gate0, stream))
                return rc;
        if (m_split < M)
            return fp_gemm_bf16(a1, lda, nullptr, 0, w1, ldw, bias1, c1, ldc, M - m_split, N, K, epilogue, gelu_from, r1, ldr, gate1, stream);
        return 0;
    }
    ConvGeom cg{};
    cg.gm = K >= 8192 ? -3 : (N >= 16384 ? 2 : 8);
    const Rows2 r2{w1, bias1, gate1, m_split};
    hipStream_t st = (hipStream_t)stream;
#define FP_CASE(E)                                                                                                       \
    case E:                                                                                                              \
        return launch_w4<bf16, E, 0, bf16, 8, 0, 0, 1>(a, lda, a2_delta, k1, w0, ldw, bias0, c, ldc, M, N, K, gelu_from, \
                                                       resid, ldr, gate0, cg, st, nullptr, nullptr, QkNormArgs{}, r2)
    switch (epilogue) {
        FP_CASE(0);
        FP_CASE(1);
        default: FP_CASE(3);
    }
#undef FP_CASE
}

// The q | k | v projection of a double-stream block (epilogue 0) or the fused in-projection of a single-stream block
// (epilogue 2: GELU from column gelu_from on) WITH K2 in its epilogue: per-head RMSNorm of the q and k columns and RoPE on
// rows < rope_rows (gemm_w4.h, QkNormArgs) - bit-identical to fp_gemm_bf16 followed by fp_qk_rmsnorm_rope_bf16 on the same
// buffers, one pass over q and k less.  q / k occupy output columns [qcol0, qcol0 + heads*128) / [kcol0, ...), both
// multiples of 128.  Needs the 4-wave kernel's alignment (c 16-byte aligned, ldc % 8 == 0).
extern "C" int fp_gemm_bf16_qknorm(const void* a, int64_t lda, const void* w, int64_t ldw, const void* bias, void* c,
                                   int64_t ldc, int M, int N, int K, int epilogue, int gelu_from, const void* wq,
                                   const void* wk, const void* wq_txt, const void* wk_txt, const float* rope,
                                   int rope_rows, int qcol0, int kcol0, int heads, int head_dim, float eps,
                                   void* stream) {
    using namespace fp_gemm;
    int64_t a2_delta;
    int k1 = 0;
    if (int rc = check_dense(a, lda, nullptr, k1, w, ldw, bias, c, ldc, M, N, K, epilogue, gelu_from, nullptr, 0, nullptr,
                             a2_delta))
        return rc;
    FP_REQUIRE(epilogue == 0 || epilogue == 2);
    FP_REQUIRE(head_dim == 128 && heads > 0 && wq && wk);
    const int width = heads * 128;
    FP_REQUIRE(qcol0 >= 0 && kcol0 >= 0 && qcol0 % 128 == 0 && kcol0 % 128 == 0 && qcol0 + width <= N && kcol0 + width <= N);
    FP_REQUIRE(qcol0 + width <= kcol0 || kcol0 + width <= qcol0);
    FP_REQUIRE(rope_rows >= 0 && rope_rows <= M && (rope_rows == 0 || rope));
    FP_REQUIRE(((uintptr_t)wq & 15) == 0 && ((uintptr_t)wk & 15) == 0 && ((uintptr_t)rope & 15) == 0);
    FP_REQUIRE((!wq_txt || ((uintptr_t)wq_txt & 15) == 0) && (!wk_txt || ((uintptr_t)wk_txt & 15) == 0));
    FP_REQUIRE(((uintptr_t)c & 15) == 0 && ldc % 8 == 0);
    if (M == 0) return 0;
    ConvGeom cg{};
    cg.gm = N >= 16384 ? 2 : 8;
    QkNormArgs qk{(const bf16*)wq, (const bf16*)wk, (const bf16*)wq_txt, (const bf16*)wk_txt, rope, rope_rows, qcol0, kcol0,
                  width, eps};
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == 0)
        return launch_w4<bf16, 0, 0, bf16, 8, 1>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, nullptr, 0,
                                                 nullptr, cg, st, nullptr, nullptr, qk);
    return launch_w4<bf16, 2, 0, bf16, 8, 1>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, nullptr, 0,
                                             nullptr, cg, st, nullptr, nullptr, qk);
}

// c [M, N] fp32 = a [M, K] . w [N, K]^T + bias (fp32 [N] or NULL), bf16 operands, fp32 accumulation written out unrounded.
// The output head under `high_quality_fp32_output_for_inference` (ref hunyuan_video_packed.py:1048-1053: activations and
// proj_out's weights upcast to fp32, then nn.Linear): both operands there are bf16 VALUES in fp32 containers, a product
// of two bf16 values is exact in fp32, and the MFMA accumulates in fp32 - the same sum in another order.
extern "C" int fp_gemm_bf16_f32out(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* c,
                                   int64_t ldc, int M, int N, int K, void* stream) {
    using namespace fp_gemm;
    FP_REQUIRE(a && w && c && M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0);
    FP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && lda >= K && ldw >= K && ldc >= N);
    FP_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 15) == 0 && ((uintptr_t)bias & 15) == 0);
    FP_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + 31) / 32) < (1ll << 31));
    if (M == 0) return 0;
    const ConvGeom cg{};
    if (N <= 128)
        return launch_w4<bf16, 0, 0, float, 4>(a, lda, 0, K, w, ldw, bias, c, ldc, M, N, K, 0, nullptr, 0, nullptr, cg,
                                               (hipStream_t)stream);
    return launch_w4<bf16, 0, 0, float>(a, lda, 0, K, w, ldw, bias, c, ldc, M, N, K, 0, nullptr, 0, nullptr, cg,
                                        (hipStream_t)stream);
}
