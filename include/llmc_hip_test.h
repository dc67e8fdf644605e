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
/* The phased form K4's far update uses: C -= op(A)[:, p] B[p, :] for p = consecutive ranges of `phase_len` k (a multiple of 16),
 * one launch, bit-identical to Kd / phase_len separate llmc_test_sgemm calls. op(B) = N. */
int llmc_test_sgemm_phased(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M, int N,
                           int Kd, int TA, int phase_len, llmc_stream_t stream);
int llmc_test_gemm3(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M, int N,
                    int Kd, int TA, int epilogue, int a_upper, int b_upper, int c_upper_only, llmc_stream_t stream);
/* The k-major product (TA) of llmc_test_gemm3 in the form K3's large far updates use: both operands are first split into
 * their three bf16 planes in `ws` (6 * Kd * roundup8(max(M, N)) * 2 bytes, 16-B aligned), then multiplied by the
 * producer / MFMA-wave kernel (gemm3.hip, k_gemm3s). M % 8 == 0, N % 8 == 0, Kd a multiple of 64 and >= 128. Returns
 * LLMC_ENOTSUP when the shape is not eligible (too few tiles unless LLMC_GEMM3S_MIN_TILES lowers the bar). */
int llmc_test_gemm3_planes(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M, int N,
                           int Kd, int epilogue, int c_upper_only, void* ws, llmc_stream_t stream);
/* tools/probes/gemm3s_probe.py: the s_memtime stamps (8 waves x 128 slots, int64) one workgroup of the last k_gemm3s
 * launch wrote under LLMC_GEMM3S_DBG=4, copied to host memory. */
int llmc_test_gemm3s_stamps(long long* host_out);

#ifdef __cplusplus
}
#endif
#endif /* LLMC_HIP_TEST_H_ */
