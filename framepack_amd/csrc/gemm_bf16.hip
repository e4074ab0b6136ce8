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
