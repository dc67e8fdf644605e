// gemm3.hip — C -= A^T B for fp32 operands on the 16-bit MFMA pipe ("bf16x3").
//
// Used by K3 only (the factorisation's far updates), where results are tolerance-checked; K4 must stay on the exact
// fp32 fma chain of sgemm.hip. Every fp32 operand value is split exactly into three bf16 terms
//     a = hi + mid + lo,   hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid)
// (each term captures 8 significand bits; the residuals are exact in fp32), and a product is accumulated in fp32 as
//     a*b ~= lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi
// — six exact bf16 products; the dropped terms (mid*lo, lo*mid, lo*lo) are below 2^-24 |a||b|, the size of one
// fp32 rounding. Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each)
// per 32x32x16 block: 2.7x fewer matrix-pipe cycles for fp32-level accuracy.
//
// Layout: both operands are k-major in memory ([Kd x M], [Kd x N], row stride ld) — the shape of the Cholesky panel
// P in `T -= P^T P`. A workgroup owns a 128x128 tile; per K-step of 32 it stages both 32x128 panels through
// registers (split there) into three bf16 planes each, k-major in LDS exactly like hessian_syrk.hip's token-major
// panels (64-B units XOR-swizzled by k & 3), and feeds the MFMAs with ds_read_b64_tr_b16 transposing reads.
#include "sgemm.h"
#include "mfma_common.h"

namespace llmc {

static constexpr int G3B = 128;               // tile edge
static constexpr int G3K = 32;                // K-step
static constexpr int G3ROW = G3B * 2;         // bytes per k-row of one plane
static constexpr int G3PLANE = G3K * G3ROW;   // 8 KiB
static constexpr int G3PANEL = 3 * G3PLANE;   // hi | mid | lo

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x8 tr_frag256(LDS_AS char* p, int imm0) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm0 + 4 * G3ROW));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// split 4 fp32 values into three packed-bf16 quadruples and write them to the three planes
__device__ __forceinline__ void split_store(float4 v, LDS_AS char* plane0, int off) {
    f32x2_t a0 = {v.x, v.y}, a1 = {v.z, v.w};
    u32x2_t out[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x2_t h0 = __builtin_convertvector(a0, bf16x2_t);
        const bf16x2_t h1 = __builtin_convertvector(a1, bf16x2_t);
        out[t].x = __builtin_bit_cast(uint32_t, h0);
        out[t].y = __builtin_bit_cast(uint32_t, h1);
        if (t < 2) {
            a0 = a0 - __builtin_convertvector(h0, f32x2_t);
            a1 = a1 - __builtin_convertvector(h1, f32x2_t);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) *(LDS_AS u32x2_t*)(plane0 + t * G3PLANE + off) = out[t];
}

static constexpr int G3AROW = 80;                 // bytes per row of a row-major A plane: 32 bf16 + 16 B pad
static constexpr int G3APLANE = G3B * G3AROW;     // 10 KiB
static constexpr int G3APANEL = 3 * G3APLANE;     // 30 KiB (TA = false); k-major A uses G3PANEL = 24 KiB

// 4 consecutive k of one row -> three bf16 quadruples into the row-major A planes
__device__ __forceinline__ void split_store_rows(float4 v, LDS_AS char* plane0, int off) {
    f32x2_t a0 = {v.x, v.y}, a1 = {v.z, v.w};
    u32x2_t out[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x2_t h0 = __builtin_convertvector(a0, bf16x2_t);
        const bf16x2_t h1 = __builtin_convertvector(a1, bf16x2_t);
        out[t].x = __builtin_bit_cast(uint32_t, h0);
        out[t].y = __builtin_bit_cast(uint32_t, h1);
        if (t < 2) {
            a0 = a0 - __builtin_convertvector(h0, f32x2_t);
            a1 = a1 - __builtin_convertvector(h1, f32x2_t);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) *(LDS_AS u32x2_t*)(plane0 + t * G3APLANE + off) = out[t];
}

// TA = true : op(A)[i][k] = A[k][i], A stored [Kd x M] (k-major, transposing LDS reads)
// TA = false: op(A)[i][k] = A[i][k], A stored [M x Kd] (row-major, 16-B LDS reads of 8 consecutive k)
// B is always [Kd x N] k-major. Hints a_upper / b_upper and the epilogues are those of sgemm.h; batch via blockIdx.z.
template <bool TA>
__global__ __launch_bounds__(256, 2) void k_gemm3(SgemmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[(TA ? G3PANEL : G3APANEL) + G3PANEL];
    LDS_AS char* lds = (LDS_AS char*)smem;
    LDS_AS char* ldsB = lds + (TA ? G3PANEL : G3APANEL);
    const int z = blockIdx.z;
    const float* A = a.A + (int64_t)z * a.sA;
    const float* B = a.B + (int64_t)z * a.sB;
    float* C = a.C + (int64_t)z * a.sC;
    const bool last = z == a.batch - 1;
    const int M = last ? a.M_last : a.M, N = last ? a.N_last : a.N, Kd = last ? a.Kd_last : a.Kd;
    const int bx = a.b_upper ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int i0 = blockIdx.y * G3B, j0 = bx * G3B;
    if (i0 >= M || j0 >= N) return;
    if (a.c_upper_only && j0 + G3B <= i0) return;
    int kb = 0, ke = Kd;
    if (a.a_upper) kb = (i0 / G3K) * G3K;
    if (a.a_lower) ke = min(ke, i0 + G3B);
    if (a.b_upper) ke = min(ke, j0 + G3B);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;

    // staging geometry, k-major panels: float4 q = k-row (tid >> 5) + 8 q, columns 4 * (tid & 31)
    const int sk = tid >> 5;
    const int sc = 4 * (tid & 31);
    const int swr_off = ((sc >> 5) << 6) + (sc & 31) * 2;
    auto lds_off = [&](int k) { return k * G3ROW + (swr_off ^ ((k & 3) << 6)); };
    // row-major A: float4 q = row (tid >> 3) + 32 q, k = 4 * (tid & 7)
    const int ar = tid >> 3;
    const int ak = 4 * (tid & 7);

    const int p = lane & 15;
    const int trow = 8 * (lane >> 5) + (p >> 2);
    const int sub = 32 * ((lane >> 4) & 1) + 8 * (p & 3);
    int offA[2], offB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
        offA[m] = TA ? trow * G3ROW + (((2 * wm + m) ^ (p >> 2)) << 6) + sub
                     : (wm * 64 + m * 32 + (lane & 31)) * G3AROW + 16 * (lane >> 5);
#pragma unroll
    for (int n = 0; n < 2; ++n) offB[n] = trow * G3ROW + (((2 * wn + n) ^ (p >> 2)) << 6) + sub;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            const int k = k0 + sk + 8 * q;
            if (k < ke && j0 + sc < N) vb = *reinterpret_cast<const float4*>(B + (int64_t)k * a.ldb + j0 + sc);
            if (TA) {
                if (k < ke && i0 + sc < M) va = *reinterpret_cast<const float4*>(A + (int64_t)k * a.lda + i0 + sc);
            } else {
                const int row = i0 + ar + 32 * q, kk = k0 + ak;
                if (row < M) {
                    const float* pa = A + (int64_t)row * a.lda + kk;
                    if (kk + 3 < ke) va = *reinterpret_cast<const float4*>(pa);
                    else {
                        if (kk < ke) va.x = pa[0];
                        if (kk + 1 < ke) va.y = pa[1];
                        if (kk + 2 < ke) va.z = pa[2];
                    }
                }
            }
            ra[q] = va;
            rb[q] = vb;
        }
    };
    if (kb < ke) gload(kb);
    for (int k0 = kb; k0 < ke; k0 += G3K) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (TA) split_store(ra[q], lds, lds_off(sk + 8 * q));
            else split_store_rows(ra[q], lds, (ar + 32 * q) * G3AROW + ak * 2);
            split_store(rb[q], ldsB, lds_off(sk + 8 * q));
        }
        __syncthreads();
        if (k0 + G3K < ke) gload(k0 + G3K);
#pragma unroll
        for (int kk = 0; kk < G3K / 16; ++kk) {
            s16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (TA) fa[m][t] = tr_frag256(lds + t * G3PLANE + offA[m], kk * 16 * G3ROW);
                    else fa[m][t] = *(LDS_AS s16x8*)(lds + t * G3APLANE + offA[m] + kk * 32);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) fb[n][t] = tr_frag256(ldsB + t * G3PLANE + offB[n], kk * 16 * G3ROW);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x16 c = acc[m][n];
                    c = Mfma<LLMC_BF16>::run(fa[m][2], fb[n][0], c);   // lo  * hi
                    c = Mfma<LLMC_BF16>::run(fa[m][0], fb[n][2], c);   // hi  * lo
                    c = Mfma<LLMC_BF16>::run(fa[m][1], fb[n][1], c);   // mid * mid
                    c = Mfma<LLMC_BF16>::run(fa[m][1], fb[n][0], c);   // mid * hi
                    c = Mfma<LLMC_BF16>::run(fa[m][0], fb[n][1], c);   // hi  * mid
                    c = Mfma<LLMC_BF16>::run(fa[m][0], fb[n][0], c);   // hi  * hi
                    acc[m][n] = c;
                }
        }
        __syncthreads();   // every wave has read this stage before it is overwritten
    }

    // epilogue; C -= acc with 16 loads in flight, then 16 stores
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = j0 + wn * 64 + n * 32 + (lane & 31);
            const bool colok = col < N;
            float old[16];
            if (a.epilogue == SG_SUB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    old[r] = (colok && row < M) ? C[(int64_t)row * a.ldc + col] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (!colok || row >= M) continue;
                float* pc = C + (int64_t)row * a.ldc + col;
                const float v = acc[m][n][r];
                if (a.epilogue == SG_SUB) *pc = old[r] - v;
                else if (a.epilogue == SG_SET) *pc = v;
                else *pc = -v;
            }
        }
}

int gemm3_launch(const SgemmArgs& a, bool TA, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return LLMC_OK;
    LLMC_REQUIRE(a.phase_len == 0, "gemm3: no phased mode");
    LLMC_REQUIRE((a.lda % 4 == 0) && (a.ldb % 4 == 0) && (a.N % 4 == 0) && (a.N_last % 4 == 0) &&
                     (!TA || (a.M % 4 == 0 && a.M_last % 4 == 0)) && (((uintptr_t)a.A & 15) == 0) &&
                     (((uintptr_t)a.B & 15) == 0) && (a.sA % 4 == 0) && (a.sB % 4 == 0),
                 "gemm3: operands must be 16-B aligned with ld and sizes multiples of 4");
    LLMC_REQUIRE((const void*)a.C != (const void*)a.B && (const void*)a.C != (const void*)a.A, "gemm3: no in-place product");
    dim3 grid((a.N + G3B - 1) / G3B, (a.M + G3B - 1) / G3B, a.batch);
    if (TA) hipLaunchKernelGGL((k_gemm3<true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm3<false>), grid, dim3(256), 0, st, a);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
int gemm3_tn_launch(const SgemmArgs& a, hipStream_t st) { return gemm3_launch(a, true, st); }

}  // namespace llmc

// C ABI test hook (tests/test_gptq_gpu.py): C (op) op(A) B with three bf16 terms per fp32 operand
extern "C" int llmc_test_gemm3(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M,
                               int N, int Kd, int TA, int epilogue, int a_upper, int b_upper, int c_upper_only,
                               llmc_stream_t stream) {
    llmc::SgemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = a.M_last = M; a.N = a.N_last = N; a.Kd = a.Kd_last = Kd;
    a.epilogue = epilogue; a.a_upper = a_upper; a.b_upper = b_upper; a.c_upper_only = c_upper_only; a.batch = 1;
    return llmc::gemm3_launch(a, TA != 0, (hipStream_t)stream);
}
