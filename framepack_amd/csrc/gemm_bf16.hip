// K5: the nn.Linear of every DiT block, bf16 (kernel: gemm_pipeline.h).
#include "gemm_pipeline.h"

extern "C" int fp_gemm_bf16(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw,
                            const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue,
                            int gelu_from, const void* resid, int64_t ldr, const void* gate, void* stream) {
    using namespace fp_gemm;
    int64_t a2_delta;
    if (int rc = check_dense(a, lda, a2, k1, w, ldw, bias, c, ldc, M, N, K, epilogue, gelu_from, resid, ldr, gate, a2_delta))
        return rc;
    if (M == 0) return 0;
    const ConvGeom cg{};
    hipStream_t st = (hipStream_t)stream;
#define FP_CASE(E) case E: return launch<bf16, E, 0>(a, lda, a2_delta, k1, w, ldw, bias, c, ldc, M, N, K, gelu_from, resid, ldr, gate, cg, st)
    switch (epilogue) {
        FP_CASE(0);
        FP_CASE(1);
        FP_CASE(2);
        FP_CASE(3);
        default: FP_CASE(4);
    }
#undef FP_CASE
}
