// sgemm.h — internal fp32 GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32), used by the GPTQ
// Cholesky / triangular inverse (K3) and the GPTQ trailing update (K4).
//
// Numerics contract (relied on by K4's bit-exact parity): every output element is ONE accumulator that
// receives its products in ascending k order, acc = fma(a_k, b_k, acc) starting from +0 — exactly the
// chain MKL's sgemm produces for the reference on CPU (tests/test_oracle_golden.py pins that) — and the
// epilogue applies C = C - acc (or = acc / = -acc) as one further rounding.
#pragma once
#include "common.h"

namespace llmc {

enum SgemmEpilogue { SG_SUB = 0 /* C -= AB */, SG_SET = 1 /* C = AB */, SG_NEG = 2 /* C = -AB */ };

struct SgemmArgs {
    const float* A;  // op(A) is [M x Kd]; stored [M x Kd] (TA=false) or [Kd x M] (TA=true), row-major, ld = lda
    const float* B;  // op(B) is [Kd x N]; stored [Kd x N] (TB=false) or [N x Kd] (TB=true)
    float* C;        // [M x N], ldc
    int64_t lda, ldb, ldc;
    int M, N, Kd;
    int epilogue;
    // structure hints (skip work that multiplies known zeros; never changes a result bit)
    int a_upper;     // op(A)[i][k] == 0 for k < i  -> start k at the tile's first row
    int a_lower;     // op(A)[i][k] == 0 for k > i  -> stop k after the tile's last row
    int b_upper;     // op(B)[k][j] == 0 for k > j  -> stop k after the tile's last column
    int c_upper_only;  // only tiles that intersect j >= i are computed/stored (symmetric update, upper half)
    // phased accumulation (SG_SUB only): every `phase_len` k (a multiple of 16) the accumulator is subtracted
    // from the C tile held in registers and reset to +0:  C -= A[:, p] B[p, :] phase by phase, i.e. exactly what
    // Kd / phase_len separate launches would compute, with one read and one write of C. 0 = single phase.
    int phase_len;
    // batch: blockIdx.z-th problem at A + z*sA etc.; dims of the LAST problem may be smaller
    int64_t sA, sB, sC;
    int batch;
    int M_last, N_last, Kd_last;
    // gemm3 only (k-major A and B): the operands already split into bf16 planes (gemm3_split_planes) — hi | mid | lo planes of
    // `plane_stride` elements each, row stride ldp, pointing at the operand's first column. nullptr = split in the kernel.
    const void* planesA;
    const void* planesB;
    int64_t ldp, plane_stride;
    int planes_dma;   // set by gemm3_launch: the planes form's producers copy by LDS-DMA (0: through registers, option gemm3s_no_dma)
};

// launches on `st`; returns LLMC_* status
int sgemm_launch(const SgemmArgs& a, bool TA, bool TB, hipStream_t st);

// K4's phased far update (TA, op(B) = N, SG_SUB, phase_len = 128, whole 256 x 128 tiles) on the one-wave-per-SIMD kernel of
// sgemm_wide.hip: same chain, same bits as sgemm_launch's other kernels
bool sgemm_wide_eligible(const SgemmArgs& a, bool TA, bool TB);
int sgemm_wide_launch(const SgemmArgs& a, hipStream_t st);

// C -= A^T B (A [Kd x M], B [Kd x N], both k-major) on the 16-bit MFMA pipe with three bf16 terms per fp32 operand
// (gemm3.hip): fp32-level accuracy, NOT the bitwise fma chain above — K3 only.
int gemm3_tn_launch(const SgemmArgs& a, hipStream_t st);
// general form: TA as in sgemm_launch, op(B) = N; hints a_upper / a_lower / b_upper / c_upper_only, all epilogues, batch
int gemm3_launch(const SgemmArgs& a, bool TA, hipStream_t st);
// the planes form on 128 x 128 tiles with two workgroups per CU (gemm3_wide.hip): SG_SUB, whole tiles, planes given
bool gemm3w_eligible(const SgemmArgs& a);
int gemm3w_launch(const SgemmArgs& a, hipStream_t st);
// true when gemm3_launch(a, true, ..) would run the planes form for these arguments (a.planesA set): split only then
bool gemm3_uses_planes(const SgemmArgs& a);
// hi | mid | lo bf16 planes of a k-major fp32 panel [rows x n] (n % 8 == 0): planes + t * plane_stride + r * ldp + c
int gemm3_split_planes(const float* P, int64_t ld, int rows, int n, void* planes, int64_t ldp, int64_t plane_stride, hipStream_t st);

// C [M, N] = sign * A B (A [M, Kd] row-major, B [Kd, N] k-major, fp32) with the same three-bf16-term arithmetic on the
// one-wave-per-SIMD GEMM of linear_eval.hip: operands split ONCE into k-tiled stacked planes. For the large, deep levels of
// K3's triangular inverse. Kd % 256 == 0; ws: gemm6_ws_bytes, 256-B aligned.
size_t gemm6_ws_bytes(int M, int N, int Kd);
int gemm6_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int Kd,
                 int a_upper, int b_upper, float sign, void* ws, hipStream_t st);

}  // namespace llmc
