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
// Two kernels: k_gemm3 (128 x 128 tiles, every wave splits and multiplies: the small and the hinted products) and, for the
// factorisation's large far updates, k_gemm3s (256 x 128 tiles, producer waves / MFMA waves, operands split once into bf16
// planes by k_split3_planes) — same terms, same MFMA order, same bits; see the comment above k_gemm3s for the measurements.
//
// Layout: both operands are k-major in memory ([Kd x M], [Kd x N], row stride ld) — the shape of the Cholesky panel
// P in `T -= P^T P`. A workgroup owns a 128x128 tile; per K-step of 32 it stages both 32x128 panels through
// registers (split there) into three bf16 planes each, k-major in LDS exactly like hessian_syrk.hip's token-major
// panels (64-B units XOR-swizzled by k & 3), and feeds the MFMAs with ds_read_b64_tr_b16 transposing reads.
#include <type_traits>
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

// ---------------------------------------------------------------------------------------------
// k_gemm3s — the same product (k-major operands, C (op) A^T B, six bf16 products per fp32 product, bit-identical C) with the
// two jobs of k_gemm3 given to DIFFERENT waves: waves 0..3 (one per SIMD) only read fragments and issue MFMAs, waves 4..7
// (their SIMD partners) only bring the three bf16 planes of the NEXT K-step into the other LDS buffer. One barrier per
// K-step.
//   PRE = false: the producers fetch the fp32 panels and split them (the arithmetic of split_store);
//   PRE = true : the panels were split ONCE into bf16 planes in memory (k_split3_planes: the same arithmetic), the producers
//                only copy. In k_gemm3 every 128 x 128 tile splits its two panels again: a panel of a 13312-column far
//                update is split 104 times.
//
// Why (profiles/r05_gemm3_pmc.txt, r05_gemm3s.txt): k_gemm3 sits at 0.38 MFMA-busy with its waves issue-stalled half of
// the time. Every wave alternates a VALU phase (the split: 5-6 VALU instructions per MFMA, 5.5 cycles each from one wave)
// with an MFMA phase, and on this chip the VALU work of a SIMD does not hide under MFMAs — not the wave's own (round 3),
// and hardly a partner wave's either: with the split in dedicated producer waves (PRE = false) a K-step still takes the
// SUM of the MFMA wave's 3072 matrix-pipe cycles and the producer's ~2000 cycles (stamps: 5600 per step). Workgroup-wide
// tilings with all waves doing both jobs (round 5's k_gemm3w, three forms) and a deeper register prefetch ended at the same
// 0.36-0.40 for the same reason. What removes the VALU work from the loop is not doing it there.
//
// Tile 256 (M) x 128 (N), K-step 32; MFMA wave (wm, wn) owns rows 128 wm .. +128, columns 64 wn .. +64 (128 accumulator
// registers); LDS: two buffers of {A: 3 planes x 32 x 512 B, B: 3 planes x 32 x 256 B} = 144 KB, one workgroup per CU.
// Panels / planes come in through range-checked buffer loads two K-steps ahead of their use. Epilogue: the accumulators go
// through LDS (the planes are dead) to the producer waves, which hold the old C values (16-B row-contiguous loads issued
// before the last K-steps finish) and write the result — from the MFMA waves' registers the same update is 8 rounds of 16
// strided 4-B loads and stores per lane, 30 us of a 78-us tile with nothing else resident on the CU.
// No operand hints (a_upper / a_lower / b_upper): the far updates have none.
// ---------------------------------------------------------------------------------------------
static constexpr int S_BM = 256, S_BN = 128;
static constexpr int S_AROW = S_BM * 2, S_BROW = S_BN * 2;                 // bytes per k-row of one plane
static constexpr int S_APLANE = G3K * S_AROW, S_BPLANE = G3K * S_BROW;     // 16 KiB, 8 KiB
static constexpr int S_BUF = 3 * S_APLANE + 3 * S_BPLANE;                  // 72 KiB
static constexpr int S_LDS = 2 * S_BUF;
// LDS-DMA form of the planes kernel (round 6): a ring of FOUR 16-k slots instead of two 32-k buffers, so that a slot's copies are
// requested three steps before its MFMAs (a DMA round trip is longer than one step's MFMAs)
static constexpr int R_K = 16;
static constexpr int R_APLANE = R_K * S_AROW, R_BPLANE = R_K * S_BROW;       // 8 KiB, 4 KiB
static constexpr int R_SLOT = 3 * R_APLANE + 3 * R_BPLANE;                    // 36 KiB
static constexpr int R_SLOTS = 4;
static_assert(R_SLOTS * R_SLOT == S_LDS, "the ring takes the two buffers' LDS");

template <int ROW>
__device__ __forceinline__ s16x8 tr_frag_row(LDS_AS char* p, int imm0) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm0 + 4 * ROW));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// the three bf16 terms of four fp32 values (split_store's arithmetic), as three packed quadruples
__device__ __forceinline__ void split3(f32x4 v, u32x2_t (&out)[3]) {
    f32x2_t a0 = {v[0], v[1]}, a1 = {v[2], v[3]};
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
}

// P [rows x n] fp32 (ld) -> planes [3][rows][ldp] bf16: hi | mid | lo of every element. n % 8 == 0; one thread per 8 columns.
__global__ __launch_bounds__(256) void k_split3_planes(const float* __restrict__ P, int64_t ld, int rows, int n,
                                                       uint16_t* __restrict__ planes, int64_t ldp, int64_t ps) {
    const int c8 = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (8 * c8 >= n) return;
    const float* src = P + (int64_t)r * ld + 8 * c8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    u32x2_t o0[3], o1[3];
    split3(v0, o0);
    split3(v1, o1);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4_t*>(planes + t * ps + (int64_t)r * ldp + 8 * c8) = u32x4_t{o0[t].x, o0[t].y, o1[t].x, o1[t].y};
    }
}

__device__ long long g_g3s_stamps[8 * 128];
#define G3S_STAMP(slot) do { if (DBG == 4 && stamp_wg && lane == 0 && (slot) < 128) g_g3s_stamps[wv * 128 + (slot)] = __builtin_readcyclecounter(); } while (0)
template <int DBG, bool PRE>   // DBG (timing experiments, tools/probes/gemm3s_probe.py): 2 = no MFMAs (wrong results), 4 = stamps
__global__ __launch_bounds__(512, 1) void k_gemm3s(SgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    LDS_AS char* lds = (LDS_AS char*)smem_s;
    const int z = blockIdx.z;
    const float* A = a.A + (int64_t)z * a.sA;
    const float* B = a.B + (int64_t)z * a.sB;
    float* C = a.C + (int64_t)z * a.sC;
    const bool last = z == a.batch - 1;
    const int M = last ? a.M_last : a.M, N = last ? a.N_last : a.N, Kd = last ? a.Kd_last : a.Kd;
    const int i0 = blockIdx.y * S_BM, j0 = blockIdx.x * S_BN;
    if (i0 >= M || j0 >= N) return;
    if (a.c_upper_only && j0 + S_BN <= i0) return;
    const int nsteps = (Kd + G3K - 1) / G3K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stamp_wg = blockIdx.x == gridDim.x - 1 && blockIdx.y == 2 && blockIdx.z == 0;
    G3S_STAMP(0);
    LDS_AS float* tile = (LDS_AS float*)lds;     // the epilogue's 256 x 128 fp32 image of the accumulators

    if constexpr (PRE && DBG == 0) {
      if (wv >= 4 && a.planes_dma) {
        // ---------------- producer waves, planes form (round 6): the planes of step j go into buffer j & 1 by LDS-DMA
        // (buffer_load_dwordx4 ... lds: 64 lanes x 16 B per instruction straight into LDS, no staging registers, no ds_write):
        // 18 instructions per wave and K-step instead of 18 loads + 18 LDS writes per THREAD. The round-5 stamps
        // (profiles/r05_gemm3s.txt) showed 7800 cycles before the first MFMA, a 5800-cycle stall at the third-last step (32 old-C
        // loads per thread issued there) and 3750-4050 cycles per 3072-cycle K-step; here the first planes land after one
        // DMA round trip, the old C values are requested four per K-step from the third step on (the staging registers that
        // held plane data are free for them), and a K-step's copy work is 18 issue slots.
        // A 1-KiB piece = two 512-B k-rows of an A plane (lanes 0-31 / 32-63) or four 256-B k-rows of a B plane (16 lanes each);
        // the LDS image is lane-linear, the rows' 64-B-unit XOR swizzle (unit ^ (k & 3), what the fragment reads undo) is
        // applied to the per-lane SOURCE column instead. Same planes, same LDS image, same MFMA order: the same bits.
        const int pw = wv - 4;
        auto mk = [](const void* p, uint32_t bytes) {     // from provably wave-uniform inputs
            const uint64_t u = (uint64_t)p;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
            return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0,
                                                     __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
        };
        const uint16_t* pA = (const uint16_t*)a.planesA + (int64_t)z * a.sA;
        const uint16_t* pB = (const uint16_t*)a.planesB + (int64_t)z * a.sB;
        const auto dA = mk(pA + i0, (uint32_t)((2 * a.plane_stride + ((int64_t)(Kd - 1)) * a.ldp + min(S_BM, M - i0)) * 2));
        const auto dB = mk(pB + j0, (uint32_t)((2 * a.plane_stride + ((int64_t)(Kd - 1)) * a.ldp + min(S_BN, N - j0)) * 2));
        // A pieces of this wave are p = pw, pw + 4, ..: k = 2 p, so (k + r) & 3 = 2 (pw & 1) + r for the lane's row r
        uint32_t voA, voB;
        {
            const int r = lane >> 5, c = lane & 31, k3 = 2 * (pw & 1) + r;
            const int ca = (((c >> 2) ^ k3) << 2) | (c & 3);
            voA = (uint32_t)(((int64_t)r * a.ldp + 8 * ca) * 2) | (i0 + 8 * ca < M ? 0u : 0x80000000u);
        }
        {
            const int r = lane >> 4, c = lane & 15;
            const int cb = (((c >> 2) ^ r) << 2) | (c & 3);
            voB = (uint32_t)(((int64_t)r * a.ldp + 8 * cb) * 2) | (j0 + 8 * cb < N ? 0u : 0x80000000u);
        }
        const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
        const uint32_t row2 = (uint32_t)(a.ldp * 2), ps2 = (uint32_t)(a.plane_stride * 2);
        auto dma = [&](const decltype(dA)& d, uint32_t vo, uint32_t so, uint32_t dst) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                         :: "v"(vo), "s"(d), "s"(dst), "s"(so) : "memory");
        };
        const int rsteps = Kd / R_K;                      // 16-k ring steps (Kd % 64 == 0, >= 128: an even number >= 8)
        auto issue = [&](int j) {
            const uint32_t slot = lds0 + (uint32_t)(j & (R_SLOTS - 1)) * R_SLOT;
            const uint32_t k0 = (uint32_t)j * R_K;
#pragma unroll
            for (int t = 0; t < 6; ++t) {                  // A: piece index 4 t + pw = 8 plane + p (two k-rows each)
                const int idx = 4 * t + pw, pl = idx >> 3, pp = idx & 7;
                dma(dA, voA, (uint32_t)pl * ps2 + (k0 + 2u * pp) * row2, slot + (uint32_t)pl * R_APLANE + (uint32_t)pp * 1024u);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {                  // B: piece index 4 t + pw = 4 plane + q (four k-rows each)
                const int idx = 4 * t + pw, pl = idx >> 2, q = idx & 3;
                dma(dB, voB, (uint32_t)pl * ps2 + (k0 + 4u * q) * row2, slot + 3u * R_APLANE + (uint32_t)pl * R_BPLANE + (uint32_t)q * 1024u);
            }
        };
        // the old C values: this thread's 32 quadruples (row (t >> 5) + 8 q, columns 4 (t & 31)), four per step behind steps 0 .. 7
        const int t = tid - 256;
        const int c4 = 4 * (t & 31), r0 = t >> 5;
        const int col = j0 + c4;
        const auto dC = mk(C + (int64_t)i0 * a.ldc + j0, (uint32_t)((((int64_t)(min(S_BM, M - i0) - 1)) * a.ldc + min(S_BN, N - j0)) * 4));
        const int row_lim = col >= N ? 0 : (a.c_upper_only ? min(M, (col & ~31) + 32) : M) - i0 - r0;
        const uint32_t c_base = (uint32_t)(((int64_t)r0 * a.ldc + c4) * 4), c_step = (uint32_t)(a.ldc * 32);
        auto c_off = [&](int q) { return (c_base + (uint32_t)q * c_step) | (8 * q < row_lim ? 0u : 0x80000000u); };
        f32x4 old[32];
        issue(0);
        issue(1);
        issue(2);
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");     // step 0's nine pieces of this wave have landed (18 younger ones may fly)
        __syncthreads();                 // step 0 is in its slot
        // Loop body j: the MFMA waves work on step j; slot (j + 3) & 3 = (j - 1) & 3 was read in step j - 1, which every MFMA wave
        // had finished at the barrier before: request step j + 3 into it, wait for step j + 1, publish it. The counted waits
        // ignore the C loads mixed in (they only make a wait stricter). Straight-line C loads (behind per-lane or per-step
        // branches hipcc's vmcnt bookkeeping cannot count them): the first eight bodies are peeled (rsteps >= 8).
        int j = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j + 3 < rsteps) issue(j + 3);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                old[4 * u + e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dC, c_off(4 * u + e), 0, 0));
            if (j + 1 < rsteps) {
                if (j + 3 < rsteps) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");        // younger than step j + 1: steps j + 2, j + 3 (18) and the C loads of three bodies (12)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            ++j;
        }
        for (; j + 1 < rsteps; ++j) {
            if (j + 3 < rsteps) {
                issue(j + 3);
                asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            } else if (j + 2 < rsteps) {
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        __syncthreads();                 // (E1) the MFMA waves are done with the planes
        __syncthreads();                 // (E2) the accumulators are in LDS
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const f32x4 v = *(LDS_AS f32x4*)(tile + (r0 + 8 * q) * S_BN + c4);
            f32x4 w;
            if (a.epilogue == SG_SUB) w = old[q] - v;
            else if (a.epilogue == SG_SET) w = v;
            else w = -v;
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w), dC, c_off(q), 0, 0);
        }
        return;
      }
    }
    if (wv >= 4) {
        // ---------------- producer waves: bring the planes of step j into buffer j & 1
        const int t = tid - 256;
        auto mk = [](const void* p, uint32_t bytes) {     // from provably wave-uniform inputs
            const uint64_t u = (uint64_t)p;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
            return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0,
                                                     __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
        };
        // NA / NB 16-B loads per thread and K-step. fp32 panels: A float4 q = k-row (t >> 6) + 4 q, columns 4 (t & 63); B: k-row
        // (t >> 5) + 8 q, columns 4 (t & 31). Planes: A 16-B chunk q of plane pl = k-row (t >> 5) + 8 q, columns 8 (t & 31);
        // B: k-row (t >> 4) + 16 q, columns 8 (t & 15).
        constexpr int NA = PRE ? 12 : 8, NB_ = PRE ? 6 : 4;
        uint32_t voA[NA], voB[NB_];
        int loA[NA], loB[NB_];
        uint32_t incA, incB;
        decltype(mk(nullptr, 0u)) dA, dB;
        if constexpr (PRE) {
            const uint16_t* pA = (const uint16_t*)a.planesA + (int64_t)z * a.sA;
            const uint16_t* pB = (const uint16_t*)a.planesB + (int64_t)z * a.sB;
            // range: up to the last valid element of row Kd - 1 of the THIRD plane
            dA = mk(pA + i0, (uint32_t)((2 * a.plane_stride + ((int64_t)(Kd - 1)) * a.ldp + min(S_BM, M - i0)) * 2));
            dB = mk(pB + j0, (uint32_t)((2 * a.plane_stride + ((int64_t)(Kd - 1)) * a.ldp + min(S_BN, N - j0)) * 2));
            const int ca = t & 31, cb = t & 15;
            const uint32_t okA = i0 + 8 * ca < M ? 0u : 0x80000000u, okB = j0 + 8 * cb < N ? 0u : 0x80000000u;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = (t >> 5) + 8 * q;
                    voA[4 * pl + q] = (uint32_t)((pl * a.plane_stride + (int64_t)k * a.ldp + 8 * ca) * 2) | okA;
                    loA[4 * pl + q] = pl * S_APLANE + k * S_AROW + ((((ca >> 2) ^ (k & 3)) << 6) + 16 * (ca & 3));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = (t >> 4) + 16 * q;
                    voB[2 * pl + q] = (uint32_t)((pl * a.plane_stride + (int64_t)k * a.ldp + 8 * cb) * 2) | okB;
                    loB[2 * pl + q] = pl * S_BPLANE + k * S_BROW + ((((cb >> 2) ^ (k & 3)) << 6) + 16 * (cb & 3));
                }
            }
            incA = incB = (uint32_t)(a.ldp * G3K * 2);
        } else {
            // range: up to the last valid element of row Kd - 1 of the tile's columns
            dA = mk(A + i0, (uint32_t)((((int64_t)(Kd - 1)) * a.lda + min(S_BM, M - i0)) * 4));
            dB = mk(B + j0, (uint32_t)((((int64_t)(Kd - 1)) * a.ldb + min(S_BN, N - j0)) * 4));
            const int ca = 4 * (t & 63), cb = 4 * (t & 31);
            const uint32_t okA = i0 + ca < M ? 0u : 0x80000000u, okB = j0 + cb < N ? 0u : 0x80000000u;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int k = (t >> 6) + 4 * q;
                voA[q] = (uint32_t)(((int64_t)k * a.lda + ca) * 4) | okA;
                loA[q] = k * S_AROW + ((((ca >> 5) ^ (k & 3)) << 6) + (ca & 31) * 2);
            }
#pragma unroll
            for (int q = 0; q < NB_; ++q) {
                const int k = (t >> 5) + 8 * q;
                voB[q] = (uint32_t)(((int64_t)k * a.ldb + cb) * 4) | okB;
                loB[q] = k * S_BROW + ((((cb >> 5) ^ (k & 3)) << 6) + (cb & 31) * 2);
            }
            incA = (uint32_t)(a.lda * G3K * 4);
            incB = (uint32_t)(a.ldb * G3K * 4);
        }
        f32x4 sa0[NA], sb0[NB_], sa1[NA], sb1[NB_];
        auto bload = [&](f32x4 (&sa)[NA], f32x4 (&sb)[NB_], int step) {
            // the step's offset is part of the per-lane offset, which the range check covers
            const uint32_t oA = (uint32_t)step * incA, oB = (uint32_t)step * incB;
#pragma unroll
            for (int q = 0; q < NA; ++q) sa[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dA, voA[q] + oA, 0, 0));
#pragma unroll
            for (int q = 0; q < NB_; ++q) sb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dB, voB[q] + oB, 0, 0));
        };
        auto produce = [&](f32x4 (&sa)[NA], f32x4 (&sb)[NB_], int j, auto reload) {
            if (DBG == 4) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NA + NB_) : "memory"); G3S_STAMP(1 + 4 * j); }
            LDS_AS char* bufA = lds + (j & 1) * S_BUF;
            LDS_AS char* bufB = bufA + 3 * S_APLANE;
            if constexpr (PRE) {
#pragma unroll
                for (int q = 0; q < NA; ++q) *(LDS_AS f32x4*)(bufA + loA[q]) = sa[q];
#pragma unroll
                for (int q = 0; q < NB_; ++q) *(LDS_AS f32x4*)(bufB + loB[q]) = sb[q];
            } else {
#pragma unroll
                for (int q = 0; q < NA; ++q) {
                    u32x2_t o[3];
                    split3(sa[q], o);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *(LDS_AS u32x2_t*)(bufA + pl * S_APLANE + loA[q]) = o[pl];
                }
#pragma unroll
                for (int q = 0; q < NB_; ++q) {
                    u32x2_t o[3];
                    split3(sb[q], o);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *(LDS_AS u32x2_t*)(bufB + pl * S_BPLANE + loB[q]) = o[pl];
                }
            }
            if constexpr (decltype(reload)::value) bload(sa, sb, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (DBG == 4) { G3S_STAMP(2 + 4 * j); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G3S_STAMP(3 + 4 * j); }
        };
        bload(sa0, sb0, 0);
        __builtin_amdgcn_sched_barrier(0);     // the two sets in issue order: a step waits for its own set only
        bload(sa1, sb1, 1);
        __builtin_amdgcn_sched_barrier(0);
        // nsteps is even and >= 4 (gemm3s_eligible): step j sits in set j & 1. The last two steps are written without a
        // reload, and the registers of the set that is free first take the old C values: this thread's 32 quadruples (row
        // (t >> 5) + 8 q, columns 4 (t & 31)) are in flight while the MFMA waves work through the last two K-steps (issued
        // after the last step they came back 10.8k cycles after the last MFMA: stamps, profiles/r05_gemm3s.txt). 32 x 32
        // blocks entirely below the diagonal of an upper-only update are neither read nor written.
        constexpr std::true_type RELOAD{};
        constexpr std::false_type LAST{};
        produce(sa0, sb0, 0, RELOAD);
        __syncthreads();
        G3S_STAMP(4);
        int j = 1;
        for (; j + 1 <= nsteps - 4; j += 2) {
            produce(sa1, sb1, j, RELOAD);
            __syncthreads();
            G3S_STAMP(4 + 4 * j);
            produce(sa0, sb0, j + 1, RELOAD);
            __syncthreads();
            G3S_STAMP(8 + 4 * j);
        }
        produce(sa1, sb1, nsteps - 3, RELOAD);
        const int c4 = 4 * (t & 31), r0 = t >> 5;
        const int col = j0 + c4;
        // Through a descriptor of the C tile, a quadruple that is not wanted at an out-of-range offset, and the loads issued
        // whatever the epilogue: straight-line code. Behind branches (per lane or per epilogue) hipcc's vmcnt bookkeeping
        // cannot count them and makes the last plane store wait for every one of them.
        const auto dC = mk(C + (int64_t)i0 * a.ldc + j0, (uint32_t)((((int64_t)(min(S_BM, M - i0) - 1)) * a.ldc + min(S_BN, N - j0)) * 4));
        // wanted rows of this thread's column quadruple: below row_lim (an upper-only update stops above the first 32 x 32
        // block that lies entirely below the diagonal; a column beyond N wants nothing). Few VALU instructions per
        // quadruple: the producers issue them beside MFMA waves that hold the issue priority.
        const int row_lim = col >= N ? 0 : (a.c_upper_only ? min(M, (col & ~31) + 32) : M) - i0 - r0;   // relative to this thread's first row
        const uint32_t c_base = (uint32_t)(((int64_t)r0 * a.ldc + c4) * 4), c_step = (uint32_t)(a.ldc * 32);
        auto c_off = [&](int q) { return (c_base + (uint32_t)q * c_step) | (8 * q < row_lim ? 0u : 0x80000000u); };
        // sixteen of them behind the third-last step, sixteen behind the second-last: 32 one-KB loads that miss the L2 take a wave
        // 7-10k cycles to ISSUE (the CU's miss handling runs at ~13 B/clk), two K-steps' worth
        f32x4 old[32];
#pragma unroll
        for (int q = 0; q < 16; ++q) old[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dC, c_off(q), 0, 0));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        G3S_STAMP(4 + 4 * (nsteps - 3));
        produce(sa0, sb0, nsteps - 2, LAST);
#pragma unroll
        for (int q = 16; q < 32; ++q) old[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dC, c_off(q), 0, 0));
        __builtin_amdgcn_sched_barrier(0);
        G3S_STAMP(100);
        if (DBG == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G3S_STAMP(101); }
        __syncthreads();
        G3S_STAMP(4 + 4 * (nsteps - 2));
        produce(sa1, sb1, nsteps - 1, LAST);
        __syncthreads();
        G3S_STAMP(4 + 4 * (nsteps - 1));
        __syncthreads();             // (E1) the MFMA waves are done with the planes
        __syncthreads();             // (E2) the accumulators are in LDS
        G3S_STAMP(120);
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const f32x4 v = *(LDS_AS f32x4*)(tile + (r0 + 8 * q) * S_BN + c4);
            f32x4 w;
            if (a.epilogue == SG_SUB) w = old[q] - v;
            else if (a.epilogue == SG_SET) w = v;
            else w = -v;
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w), dC, c_off(q), 0, 0);
        }
        if (DBG == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); G3S_STAMP(121); }
        return;
    }

    // ---------------- MFMA waves
    __builtin_amdgcn_s_setprio(3);   // the matrix pipe's wave wins the issue arbitration (849 -> 801 us on a 13312^2 x 512 update)
    const int wm = wv >> 1, wn = wv & 1;
    const int p = lane & 15;
    const int trow = 8 * (lane >> 5) + (p >> 2);
    const int sub = 32 * ((lane >> 4) & 1) + 8 * (p & 3);
    int offA[4], offB[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) offA[m] = trow * S_AROW + (((4 * wm + m) ^ (p >> 2)) << 6) + sub;
#pragma unroll
    for (int n = 0; n < 2; ++n) offB[n] = trow * S_BROW + (((2 * wn + n) ^ (p >> 2)) << 6) + sub;

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    __syncthreads();                 // step 0 is in buffer 0
    if constexpr (DBG == 0) {
        // Round 6: the fragment reads of an accumulator row are issued one row AHEAD of its MFMAs (A fragments double-buffered per
        // row, B fragments per 16-k slice: 72 fragment registers as before plus 36), every phase pinned by a scheduling barrier.
        // The compiler's own schedule read a row's fragments and then waited for them in front of its MFMAs — 3750-4050 cycles
        // per 3072-cycle K-step in the round-5 stamps. Per accumulator the six products keep k_gemm3's order: the same bits.
        // TWO accumulator rows at a time: the six products of an accumulator are a dependent chain (k_gemm3's order is kept, so
        // the bits are), and with only the row's two accumulators alternating every MFMA waited ~12 cycles for the one two
        // places before it (3750-4050 cycles per 3072-cycle K-step in the round-5 stamps); four accumulators in rotation hide it.
        auto ldP = [&](s16x8 (&f)[2][3], LDS_AS char* bA, int aplane, int m0, int imm) {     // A fragments of rows m0, m0 + 1
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 2; ++r) f[r][t] = tr_frag_row<S_AROW>(bA + t * aplane + offA[m0 + r], imm);
        };
        auto ldB = [&](s16x8 (&f)[2][3], LDS_AS char* bB, int bplane, int imm) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) f[n][t] = tr_frag_row<S_BROW>(bB + t * bplane + offB[n], imm);
        };
        auto pair = [&](auto mc, const s16x8 (&fa)[2][3], const s16x8 (&fb)[2][3]) {
            constexpr int m = decltype(mc)::value;
            f32x16 c00 = acc[m][0], c01 = acc[m][1], c10 = acc[m + 1][0], c11 = acc[m + 1][1];
#define LLMC_G3_PROD(TA, TB)                                      \
            c00 = Mfma<LLMC_BF16>::run(fa[0][TA], fb[0][TB], c00);    \
            c01 = Mfma<LLMC_BF16>::run(fa[0][TA], fb[1][TB], c01);    \
            c10 = Mfma<LLMC_BF16>::run(fa[1][TA], fb[0][TB], c10);    \
            c11 = Mfma<LLMC_BF16>::run(fa[1][TA], fb[1][TB], c11);
            LLMC_G3_PROD(2, 0)   // lo  * hi   (k_gemm3's order per accumulator)
            LLMC_G3_PROD(0, 2)   // hi  * lo
            LLMC_G3_PROD(1, 1)   // mid * mid
            LLMC_G3_PROD(1, 0)   // mid * hi
            LLMC_G3_PROD(0, 1)   // hi  * mid
            LLMC_G3_PROD(0, 0)   // hi  * hi
#undef LLMC_G3_PROD
            acc[m][0] = c00; acc[m][1] = c01; acc[m + 1][0] = c10; acc[m + 1][1] = c11;
            __builtin_amdgcn_sched_barrier(0);
        };
        constexpr std::integral_constant<int, 0> P0{};
        constexpr std::integral_constant<int, 2> P2{};
        s16x8 fb[2][3], faP[2][3], faQ[2][3];
        if (PRE && a.planes_dma) {
            // the LDS-DMA producers' ring: 16-k steps in four slots (see R_K)
            const int rsteps = Kd / R_K;
            for (int s = 0; s < rsteps; ++s) {
                LDS_AS char* bufA = lds + (s & (R_SLOTS - 1)) * R_SLOT;
                LDS_AS char* bufB = bufA + 3 * R_APLANE;
                ldB(fb, bufB, R_BPLANE, 0);
                ldP(faP, bufA, R_APLANE, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ldP(faQ, bufA, R_APLANE, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                pair(P0, faP, fb);
                pair(P2, faQ, fb);
                if (s + 1 < rsteps) __syncthreads();   // step s + 1 is in its slot; slot s & 3 may be refilled
            }
        } else {
            for (int s = 0; s < nsteps; ++s) {
                LDS_AS char* bufA = lds + (s & 1) * S_BUF;
                LDS_AS char* bufB = bufA + 3 * S_APLANE;
#pragma unroll
                for (int kk = 0; kk < G3K / 16; ++kk) {
                    ldB(fb, bufB, S_BPLANE, kk * 16 * S_BROW);
                    ldP(faP, bufA, S_APLANE, 0, kk * 16 * S_AROW);
                    __builtin_amdgcn_sched_barrier(0);
                    ldP(faQ, bufA, S_APLANE, 2, kk * 16 * S_AROW);
                    __builtin_amdgcn_sched_barrier(0);
                    pair(P0, faP, fb);
                    pair(P2, faQ, fb);
                }
                if (s + 1 < nsteps) __syncthreads();   // buffer (s + 1) & 1 is written, buffer s & 1 is free again
            }
        }
    } else
    for (int s = 0; s < nsteps; ++s) {
        G3S_STAMP(1 + 4 * s);
        LDS_AS char* bufA = lds + (s & 1) * S_BUF;
        LDS_AS char* bufB = bufA + 3 * S_APLANE;
#pragma unroll
        for (int kk = 0; kk < G3K / 16; ++kk) {
            s16x8 fa[4][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int m = 0; m < 4; ++m) fa[m][t] = tr_frag_row<S_AROW>(bufA + t * S_APLANE + offA[m], kk * 16 * S_AROW);
#pragma unroll
                for (int n = 0; n < 2; ++n) fb[n][t] = tr_frag_row<S_BROW>(bufB + t * S_BPLANE + offB[n], kk * 16 * S_BROW);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x16 c = acc[m][n];
                    if (DBG == 2) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) c[t] += (float)(fa[m][t][0] + fb[n][t][0]);
                        acc[m][n] = c;
                        continue;
                    }
                    c = Mfma<LLMC_BF16>::run(fa[m][2], fb[n][0], c);   // lo  * hi   (k_gemm3's order)
                    c = Mfma<LLMC_BF16>::run(fa[m][0], fb[n][2], c);   // hi  * lo
                    c = Mfma<LLMC_BF16>::run(fa[m][1], fb[n][1], c);   // mid * mid
                    c = Mfma<LLMC_BF16>::run(fa[m][1], fb[n][0], c);   // mid * hi
                    c = Mfma<LLMC_BF16>::run(fa[m][0], fb[n][1], c);   // hi  * mid
                    c = Mfma<LLMC_BF16>::run(fa[m][0], fb[n][0], c);   // hi  * hi
                    acc[m][n] = c;
                }
        }
        if (DBG == 4) { __builtin_amdgcn_sched_barrier(0); G3S_STAMP(2 + 4 * s); }
        if (s + 1 < nsteps) __syncthreads();   // buffer (s + 1) & 1 is written, buffer s & 1 is free again
    }
    __syncthreads();                 // (E1) every MFMA wave has read its last fragments
    G3S_STAMP(120);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                tile[row * S_BN + wn * 64 + n * 32 + (lane & 31)] = acc[m][n][r];
            }
    __syncthreads();                 // (E2) the tile is in LDS; the producer waves apply it
    G3S_STAMP(121);
}

// the specialised kernel for large k-major products without operand hints; everything else stays on k_gemm3
static bool gemm3s_eligible(const SgemmArgs& a) {
    if (opt(OPT_GEMM3_NOSPEC)) return false;
    if (a.a_upper || a.a_lower || a.b_upper) return false;
    if ((((uintptr_t)a.C) & 15) || a.ldc % 4 || a.sC % 4) return false;            // 16-B accesses to C
    if ((int64_t)S_BM * a.ldc * 4 >= (int64_t)0x7fffff00) return false;             // 32-bit offsets inside a C tile
    const int64_t kd = a.Kd > a.Kd_last ? a.Kd : a.Kd_last;
    const bool pre = a.planesA != nullptr;
    if (a.Kd % (2 * G3K) || a.Kd_last % (2 * G3K) || a.Kd < 4 * G3K || a.Kd_last < 4 * G3K) return false;   // an even number of K-steps, >= 4
    if (pre) {
        if (a.batch != 1) return false;                                             // the planes of ONE panel
        if (a.ldp % 8 || a.plane_stride % 8 || (((uintptr_t)a.planesA | (uintptr_t)a.planesB) & 15)) return false;
        if ((2 * a.plane_stride + (kd + G3K) * a.ldp) * 2 >= (int64_t)0x7fffff00) return false;
    } else {
        const int64_t ld = a.lda > a.ldb ? a.lda : a.ldb;
        if ((kd + G3K) * ld * 4 >= (int64_t)0x7fffff00) return false;   // 32-bit offsets from the tile's first element
    }
    // one workgroup per CU: worth it once the tiles that do work come near filling the chip
    const int64_t tm = (a.M + S_BM - 1) / S_BM, tn = (a.N + S_BN - 1) / S_BN;
    const int64_t tiles = a.c_upper_only ? tm * tn - tm * (tm - 1) : tm * tn;      // row r of tiles skips its first 2r columns
    const int mt = opt(OPT_GEMM3S_MIN_TILES);               // the tests lower it to reach the kernel with small shapes
    const int min_tiles = mt > 0 ? mt : (pre ? 48 : 256);   // bench: 48 -> 93.75, 160 -> 93.98 / 94.20, 600 -> 94.48, never -> 95.14 ms per step
    return tiles * a.batch >= min_tiles;
}

bool gemm3_uses_planes(const SgemmArgs& a) { return a.planesA != nullptr && gemm3s_eligible(a); }

// planes of a k-major fp32 panel (rows x n, n % 8 == 0) for the PRE form: 3 * rows * ldp bf16 at `planes`
int gemm3_split_planes(const float* P, int64_t ld, int rows, int n, void* planes, int64_t ldp, int64_t plane_stride, hipStream_t st) {
    LLMC_REQUIRE(n % 8 == 0 && ld % 4 == 0 && ldp % 8 == 0 && plane_stride % 8 == 0 && (((uintptr_t)P | (uintptr_t)planes) & 15) == 0,
                 "gemm3_split_planes: 8-column granularity, 16-B aligned");
    if (rows <= 0 || n <= 0) return LLMC_OK;
    hipLaunchKernelGGL(k_split3_planes, dim3((n / 8 + 255) / 256, rows), dim3(256), 0, st, P, ld, rows, n, (uint16_t*)planes, ldp,
                       plane_stride);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

int gemm3_launch(const SgemmArgs& a, bool TA, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return LLMC_OK;
    LLMC_REQUIRE(a.phase_len == 0, "gemm3: no phased mode");
    LLMC_REQUIRE((a.lda % 4 == 0) && (a.ldb % 4 == 0) && (a.N % 4 == 0) && (a.N_last % 4 == 0) &&
                     (!TA || (a.M % 4 == 0 && a.M_last % 4 == 0)) && (((uintptr_t)a.A & 15) == 0) &&
                     (((uintptr_t)a.B & 15) == 0) && (a.sA % 4 == 0) && (a.sB % 4 == 0),
                 "gemm3: operands must be 16-B aligned with ld and sizes multiples of 4");
    LLMC_REQUIRE((const void*)a.C != (const void*)a.B && (const void*)a.C != (const void*)a.A, "gemm3: no in-place product");
    if (TA && gemm3s_eligible(a) && gemm3w_eligible(a)) return gemm3w_launch(a, st);
    if (TA && gemm3s_eligible(a)) {
        SgemmArgs b = a;
        b.planes_dma = opt(OPT_GEMM3S_NO_DMA) ? 0 : 1;
        dim3 sgrid((a.N + S_BN - 1) / S_BN, (a.M + S_BM - 1) / S_BM, a.batch);
#ifdef LLMC_LAB
        const char* dbg = lab_env("LLMC_GEMM3S_DBG");
        const int d = dbg ? atoi(dbg) : 0;
#else
        const int d = 0;
#endif
#define LLMC_G3S(D, PRE) do { if (int rc = ensure_dynamic_lds((const void*)k_gemm3s<D, PRE>, S_LDS)) return rc; \
                              hipLaunchKernelGGL((k_gemm3s<D, PRE>), sgrid, dim3(512), S_LDS, st, b); } while (0)
#ifdef LLMC_LAB
        if (a.planesA) { if (d == 2) LLMC_G3S(2, true); else if (d == 4) LLMC_G3S(4, true); else LLMC_G3S(0, true); }
        else { if (d == 2) LLMC_G3S(2, false); else if (d == 4) LLMC_G3S(4, false); else LLMC_G3S(0, false); }
#else
        (void)d;
        if (a.planesA) LLMC_G3S(0, true); else LLMC_G3S(0, false);
#endif
#undef LLMC_G3S
        LLMC_LAUNCH_CHECK();
        return LLMC_OK;
    }
    dim3 grid((a.N + G3B - 1) / G3B, (a.M + G3B - 1) / G3B, a.batch);
    if (TA) hipLaunchKernelGGL((k_gemm3<true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm3<false>), grid, dim3(256), 0, st, a);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
int gemm3_tn_launch(const SgemmArgs& a, hipStream_t st) { return gemm3_launch(a, true, st); }

}  // namespace llmc

extern "C" int llmc_test_gemm3s_stamps(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(llmc::g_g3s_stamps), sizeof(long long) * 8 * 128) == hipSuccess ? 0 : 1;
}

// C ABI test hook: the k-major product with both operands split into planes first (ws: 6 * Kd * roundup8(max(M, N)) * 2 bytes)
extern "C" int llmc_test_gemm3_planes(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc, int M,
                                      int N, int Kd, int epilogue, int c_upper_only, void* ws, llmc_stream_t stream) {
    llmc::SgemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = a.M_last = M; a.N = a.N_last = N; a.Kd = a.Kd_last = Kd;
    a.epilogue = epilogue; a.c_upper_only = c_upper_only; a.batch = 1;
    const int64_t ldp = ((M > N ? M : N) + 7) / 8 * 8, ps = (int64_t)Kd * ldp;
    a.ldp = ldp; a.plane_stride = ps;
    a.planesA = ws;
    a.planesB = (const char*)ws + 3 * ps * 2;
    if (!llmc::gemm3_uses_planes(a)) return LLMC_ENOTSUP;
    if (int rc = llmc::gemm3_split_planes(A, lda, Kd, M, ws, ldp, ps, (hipStream_t)stream)) return rc;
    if (int rc = llmc::gemm3_split_planes(B, ldb, Kd, N, (char*)ws + 3 * ps * 2, ldp, ps, (hipStream_t)stream)) return rc;
    return llmc::gemm3_launch(a, true, (hipStream_t)stream);
}

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
