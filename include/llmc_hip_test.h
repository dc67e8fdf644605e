/* llmc_hip_test.h — TEST-ONLY entry points of libllmc_hip.so (VERDICT r03 hygiene: not part of the product surface that
 * include/llmc_hip.h declares). tests/test_gptq_gpu.py and tools/bench_sgemm.py exercise the two internal GEMMs of K3 / K4
 * through them; nothing in llmc_amd/ calls them. */
#ifndef LLMC_HIP_TEST_H_
#define LLMC_HIP_TEST_H_
#include "llmc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Test hooks (tests/test_gptq_gpu.py only): the internal fp32-MFMA GEMM used by K3/K4, and the split-bf16 GEMM
 * (C (op) op(A) B, TA / epilogue / hints as in llmc_test_sgemm, three bf16 terms per operand, six products) used by K3.
 * C (op) op(A)[M x Kd] . op(B)[Kd x N]; epilogue 0: C -= AB, 1: C = AB, 2: C = -AB.
 * ---------------------------------------------------------------------------------------------- */
int llmc_test_sgemm(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M,
                    int N, int Kd, int TA, int TB, int epilogue, int a_upper, int a_lower, int b_upper,
                    int c_upper_only, llmc_stream_t stream);
int llmc_test_gemm3(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M, int N,
                    int Kd, int TA, int epilogue, int a_upper, int b_upper, int c_upper_only, llmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LLMC_HIP_TEST_H_ */
