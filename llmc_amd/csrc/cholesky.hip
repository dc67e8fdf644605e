// cholesky.hip — K2/K3 of the GPTQ path (gptq.py:139-174):
//   llmc_hessian_prep     dead-column fix, actorder gather of H and W, damping
//   llmc_chol_inv_upper   U upper with H^-1 = U^T U, the factor the reference obtains from
//                         cholesky -> cholesky_inverse -> cholesky(upper)
//
// Math (fp32 throughout, like the reference): with J the index reversal, U = J * inv(chol_lower(J H J)) * J.
// In upper/row-major form: A' = antitranspose(H); A' = U'^T U' (blocked right-looking, 128-wide panels:
// one workgroup factors + inverts the diagonal block, the panel solve and the symmetric trailing update
// are fp32-MFMA GEMMs); V = U'^-1 by recursive doubling over the already inverted diagonal blocks
// ([[A,C],[0,B]]^-1 = [[A^-1, -A^-1 C B^-1],[0, B^-1]]); U = antitranspose(V).  2K^3/3 flops instead of
// the reference's 4K^3/3.
#include <stdlib.h>
#include "common.h"
#include "sgemm.h"
#include "side_stream.h"

namespace llmc {

static constexpr int NB = 128;

// ---------------------------------------------------------------------------------------------
// out[i][j] = in[n-1-j][n-1-i]  (reflection across the anti-diagonal), 32x32 LDS tiles, coalesced both
// ways. upper_only: write 0 for j < i.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_antitranspose(const float* __restrict__ in, float* __restrict__ out,
                                                       int n, int upper_only) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;    // input tile
    for (int y = ty; y < 32; y += 8) {
        int r = r0 + y, c = c0 + tx;
        tile[y][tx] = (r < n && c < n) ? in[(int64_t)r * n + c] : 0.0f;
    }
    __syncthreads();
    // in[r][c] -> out[n-1-c][n-1-r]; output row = n-1-(c0+y'), output col = n-1-(r0+x')
    for (int y = ty; y < 32; y += 8) {
        int c = c0 + y, r = r0 + tx;
        if (r < n && c < n) {
            int orow = n - 1 - c, ocol = n - 1 - r;
            float v = tile[tx][y];
            if (upper_only && ocol < orow) v = 0.0f;
            out[(int64_t)orow * n + ocol] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Diagonal block (<= 128 x 128): U = chol_upper(D), V = U^-1, one workgroup of 4 waves, everything in LDS.
// Blocked over 32 x 32 sub-blocks:
//   - a sub-block is factored / inverted by ONE wave with lane j owning column j in 32 registers; pivots
//     and multipliers travel by v_readlane (uniform lane index), no LDS, no barriers inside;
//   - panel solves, trailing updates and the doubling steps of the inverse are 32x32x32 block products on
//     the f32 MFMA pipe, spread over the 4 waves.
// Replaces a 128-step barrier-synchronised kernel (396 us on MI355X, profiles/r01_a) on the critical path
// of the blocked factorisation. Partial blocks are padded with the identity.
// ---------------------------------------------------------------------------------------------
static constexpr int PLD = 132;  // LDS leading dimension (floats)
typedef __attribute__((ext_vector_type(16))) float pf32x16;

__device__ __forceinline__ float rdlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// wave-level: factor the 32x32 block at S (upper Cholesky, in place, strict lower zeroed) and write its inverse to V.
// An identity carried next to the block receives the SAME row operations (scale row i by 1/u_ii, subtract u_ik * row i
// from row k): the operations that turn A into U turn I into U^-T, so the inverse needs no back-substitution pass.
// 1/sqrt(d) is the hardware rsq (1 ulp) + one Newton step instead of the correctly rounded sqrt and divide (about 30
// dependent instructions on the pivot chain); nothing downstream needs those last bits. (Earlier variants kept the
// block in 32 registers per lane with v_readlane / LDS-broadcast multipliers; see the git history.)
//
// On the matrix pipe: the 32x32 block and the identity live in two MFMA accumulators
// (C layout: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31), and the rank-1 update of pivot i,
// rows k > i of [A | I] -= u_ik * (scaled row i), is ONE v_mfma_f32_32x32x2_f32 per accumulator: the scaled row sits in
// the 32 lanes of one half-wave, which is exactly where the MFMA takes both its row-indexed multipliers (A operand,
// masked to rows > i) and its column-indexed row vector (B operand) for k-slot `half`; the other k-slot is fed zeros.
// No cross-lane traffic besides the pivot's v_readlane; 2 x 64 matrix-pipe cycles per pivot instead of ~2 x 31
// readlane + fma pairs.
__device__ __forceinline__ void wave_potrf32_inv(float* __restrict__ S, float* __restrict__ V,
                                                 float* __restrict__ rowbuf, int lane, int kglobal,
                                                 int* __restrict__ info) {
#pragma clang fp contract(fast)
    (void)rowbuf;
    typedef __attribute__((ext_vector_type(16))) float acc16;
    const int j = lane & 31, h = lane >> 5;
    acc16 A, E;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        A[r] = S[row * PLD + j];
        E[r] = row == j ? 1.0f : 0.0f;
    }
    bool bad = false;
    // Round 4: pivots in PANELS of four. Rows 4q .. 4q+3 sit in the four registers 4(q>>1) .. +3 of ONE half-wave
    // (q & 1), so a panel is factored on the VALU alone — pivot, rsq + Newton, scale the row, and update the (at most
    // three) later rows of the panel with one v_readlane + one fma each, the same fma(-u, row_i, row_k) the matrix pipe
    // would do — while the matrix pipe applies the PREVIOUS panel's pivots to the identity half (E). The rows below the
    // panel then get the panel's four pivots as two rank-2 updates per accumulator (both k-slots of v_mfma_f32_32x32x2:
    // the second pivot's row is moved to the other half-wave by v_permlane32_swap). Before (round 3) every pivot waited
    // for its own two dependent MFMAs (2 x 64 pipe cycles + the accumulator read-back): 269 cycles per pivot measured
    // (profiles/r04_potrf_stamps.txt); now the matrix pipe carries 4 MFMAs per panel and the pivot chain is VALU-only.
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rb = 4 * (q >> 1), hq = q & 1;
        const bool mine = h == hq;
        float row[4], va[4], rinv[4], m[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) row[t] = A[rb + t];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * q + t;
            const float d = rdlane(row[t], hq * 32 + i);
            if (!(d > 0.0f)) bad = true;
            float ri = __builtin_amdgcn_rsqf(d);
            ri = ri * (1.5f - 0.5f * d * ri * ri);
            rinv[t] = ri;
            va[t] = (mine && j >= i) ? row[t] * ri : 0.0f;          // scaled row i (upper part); the other half-wave holds 0
#pragma unroll
            for (int s2 = t + 1; s2 < 4; ++s2) {
                m[t][s2] = rdlane(va[t], hq * 32 + 4 * q + s2);     // u_{i, 4q+s2}
                row[s2] = __builtin_fmaf(-m[t][s2], va[t], row[s2]);
            }
        }
        // the identity half of the panel, with the multipliers found above
        float er[4], ve[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) er[t] = E[rb + t];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ve[t] = mine ? er[t] * rinv[t] : 0.0f;
#pragma unroll
            for (int s2 = t + 1; s2 < 4; ++s2) er[s2] = __builtin_fmaf(-m[t][s2], ve[t], er[s2]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            A[rb + t] = mine ? va[t] : A[rb + t];
            E[rb + t] = mine ? ve[t] : E[rb + t];
        }
        if (q < 7) {
            // rows >= 4q+4: two rank-2 updates per accumulator. Operand lanes 0..31 = k-slot 0, 32..63 = k-slot 1.
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const float b0 = va[2 * pr], b1 = va[2 * pr + 1];
                const float e0 = ve[2 * pr], e1 = ve[2 * pr + 1];
                const float a0 = (mine && j >= 4 * q + 4) ? -b0 : 0.0f;      // -u_{i,k} as the multiplier of row k = j
                const float a1 = (mine && j >= 4 * q + 4) ? -b1 : 0.0f;
                float opa, opb, ope;
                if (hq == 0) {       // data in lanes 0..31: x = [p0 | 0], y = [p1 | 0] -> [p0 | p1]
                    opa = __int_as_float(__builtin_amdgcn_permlane32_swap(__float_as_int(a0), __float_as_int(a1), false, false)[0]);
                    opb = __int_as_float(__builtin_amdgcn_permlane32_swap(__float_as_int(b0), __float_as_int(b1), false, false)[0]);
                    ope = __int_as_float(__builtin_amdgcn_permlane32_swap(__float_as_int(e0), __float_as_int(e1), false, false)[0]);
                } else {             // data in lanes 32..63: swap(y, x) leaves [p1 | p0] in the second result
                    opa = __int_as_float(__builtin_amdgcn_permlane32_swap(__float_as_int(a1), __float_as_int(a0), false, false)[1]);
                    opb = __int_as_float(__builtin_amdgcn_permlane32_swap(__float_as_int(b1), __float_as_int(b0), false, false)[1]);
                    ope = __int_as_float(__builtin_amdgcn_permlane32_swap(__float_as_int(e1), __float_as_int(e0), false, false)[1]);
                }
                A = __builtin_amdgcn_mfma_f32_32x32x2f32(opa, opb, A, 0, 0, 0);
                E = __builtin_amdgcn_mfma_f32_32x32x2f32(opa, ope, E, 0, 0, 0);
            }
        }
    }
    if (bad && lane == 0) atomicCAS(info, 0, kglobal + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        S[row * PLD + j] = j >= row ? A[r] : 0.0f;
    }
    // E = U^-T: E[row][j] = V[j][row]; four consecutive rows per register quad -> one 16-B store
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(V + j * PLD + 8 * q + 4 * h) = make_float4(E[4 * q], E[4 * q + 1], E[4 * q + 2], E[4 * q + 3]);
}

// 32x32x32 block product on the f32 MFMA: acc += op(A) * B, op(A)[i][k] = TA ? A[k][i] : A[i][k]
template <bool TA>
__device__ __forceinline__ pf32x16 blk_mm(const float* __restrict__ A, const float* __restrict__ B, pf32x16 acc,
                                          int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = 2 * kk + h;
        const float av = TA ? A[k * PLD + c] : A[c * PLD + k];
        const float bv = B[k * PLD + c];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ void blk_store(float* __restrict__ C, const pf32x16& acc, int lane, float sign) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * PLD + c] = sign * acc[r];
}
__device__ __forceinline__ void blk_sub(float* __restrict__ C, const pf32x16& acc, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* p = C + ((r & 3) + 8 * (r >> 2) + 4 * h) * PLD + c;
        *p = *p - acc[r];
    }
}
#define PBLK(M, bi, bj) ((M) + (bi) * 32 * PLD + (bj) * 32)
// Blocks of the inverse V live in the part of S the factor does not use: V(a,b), a < b, in S's block (b,a) below the
// diagonal; the four diagonal blocks V(a,a) side by side in a 32-row strip Vd. That keeps the kernel at 85 KB of
// LDS (S 66 KB + strip 16.5 KB) instead of 2 x 66 KB, so it can be co-resident with workgroups of the far-update
// GEMM running on the helper stream instead of waiting for a CU to drain.
#define VBLK(a, b) ((a) == (b) ? (Vd + (a) * 32) : PBLK(S, b, a))

// phase stamps for tools/probes/probe_potrf.hip (compiled out of the product build)
#ifdef LLMC_PROBE_STAMPS
__device__ long long g_potrf_stamps[32];
#define LLMC_STAMP(i) do { if (threadIdx.x == 0) g_potrf_stamps[i] = clock64(); } while (0)
#else
#define LLMC_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256, 2) void k_potrf_inv(float* __restrict__ W, int64_t ld, int k0, int nb,
                                                   float* __restrict__ Vout, int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) float plds[];
    float* S = plds;
    float* Vd = plds + NB * PLD;
    float* rowbuf = plds + NB * PLD + 32 * PLD;   // 64 floats of wave-private scratch for the 32x32 factor
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pf32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    LLMC_STAMP(0);
    // load the block with 16 independent 16-B loads per thread in flight (a scalar loop here serialises 64
    // dependent global-load latencies and was most of this kernel's time)
    {
        float4 ld4[16];
        const bool full = nb == NB && (ld % 4 == 0) && (k0 % 4 == 0);
        // the whole-block path has NO control flow between its 16 loads: with the edge path merged into the same loop the
        // compiler waited for every load before issuing the next (16 serialised round trips, 22.6k of the kernel's 93.7k
        // cycles: profiles/r04_potrf_stamps.txt)
        if (full) {
            const float* base = W + (int64_t)(k0 + (tid >> 5)) * ld + k0 + (tid & 31) * 4;
#pragma unroll
            for (int q = 0; q < 16; ++q) ld4[q] = *reinterpret_cast<const float4*>(base + (int64_t)(8 * q) * ld);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e4 = tid + 256 * q;            // float4 index: row = e4 >> 5, col4 = e4 & 31
                const int i = e4 >> 5, j = (e4 & 31) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nb) {
                    const float* p = W + (int64_t)(k0 + i) * ld + k0 + j;
                    if (j < nb) v.x = p[0];
                    if (j + 1 < nb) v.y = p[1];
                    if (j + 2 < nb) v.z = p[2];
                    if (j + 3 < nb) v.w = p[3];
                }
                ld4[q] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e4 = tid + 256 * q;
            const int i = e4 >> 5, j = (e4 & 31) * 4;
            float vv[4] = {ld4[q].x, ld4[q].y, ld4[q].z, ld4[q].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int jj = j + t;
                float x = (i == jj) ? 1.0f : 0.0f;                 // identity padding
                if (i < nb && jj < nb) x = (jj >= i) ? vv[t] : 0.0f;
                vv[t] = x;
            }
            *reinterpret_cast<float4*>(S + i * PLD + j) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
        // V needs no initialisation: every block of its upper part is written before it is read
    }
    __syncthreads();
    LLMC_STAMP(1);
    // ---- blocked upper Cholesky over 4 block rows
    for (int kb = 0; kb < 4; ++kb) {
        if (wv == 0) wave_potrf32_inv(PBLK(S, kb, kb), VBLK(kb, kb), rowbuf, lane, k0 + kb * 32, info);
        __syncthreads();
        LLMC_STAMP(2 + 2 * kb);
        {   // panel: S(kb, jb) = V_kk^T * S(kb, jb)
            const int jb = kb + 1 + wv;
            if (jb < 4) {
                pf32x16 acc = blk_mm<true>(VBLK(kb, kb), PBLK(S, kb, jb), zero, lane);
                blk_store(PBLK(S, kb, jb), acc, lane, 1.0f);
            }
        }
        __syncthreads();
        {   // trailing: S(ib, jb) -= S(kb, ib)^T * S(kb, jb), kb < ib <= jb
            int idx = 0;
            for (int ib = kb + 1; ib < 4; ++ib)
                for (int jb = ib; jb < 4; ++jb, ++idx)
                    if ((idx & 3) == wv) {
                        pf32x16 acc = blk_mm<true>(PBLK(S, kb, ib), PBLK(S, kb, jb), zero, lane);
                        blk_sub(PBLK(S, ib, jb), acc, lane);
                    }
        }
        __syncthreads();
        LLMC_STAMP(3 + 2 * kb);
    }
    // ---- U back to the work matrix (S becomes scratch afterwards)
    if (nb == NB && (ld % 4 == 0) && (k0 % 4 == 0)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e4 = tid + 256 * q;
            const int i = e4 >> 5, j = (e4 & 31) * 4;
            // diagonal blocks hold zeros below the diagonal; blocks below it are about to receive parts of V
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((j >> 5) >= (i >> 5)) s4 = *reinterpret_cast<const float4*>(S + i * PLD + j);
            *reinterpret_cast<float4*>(W + (int64_t)(k0 + i) * ld + k0 + j) = s4;
        }
    } else {
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e >> 7, j = e & 127;
            if (i < nb && j < nb) W[(int64_t)(k0 + i) * ld + k0 + j] = (j >= i) ? S[i * PLD + j] : 0.0f;
        }
    }
    LLMC_STAMP(10);
    // ---- V = U^-1 by doubling. Level 32 -> 64: pairs (0,1) and (2,3)
    if (wv < 2) {
        const int a = 2 * wv, b = a + 1;
        pf32x16 x = blk_mm<false>(VBLK(a, a), PBLK(S, a, b), zero, lane);   // X = V_aa * U_ab
        blk_store(VBLK(a, b), x, lane, 1.0f);
        pf32x16 y = blk_mm<false>(VBLK(a, b), VBLK(b, b), zero, lane);   // Y = X * V_bb
        blk_store(VBLK(a, b), y, lane, -1.0f);
    }
    __syncthreads();
    // Level 64 -> 128: X = V[0:64,0:64] * U[0:64,64:128] (into V scratch), then -X * V[64:,64:] (into S scratch)
    {
        const int r = wv >> 1, c = 2 + (wv & 1);
        pf32x16 x = blk_mm<false>(VBLK(r, r), PBLK(S, r, c), zero, lane);
        if (r == 0) x = blk_mm<false>(VBLK(0, 1), PBLK(S, 1, c), x, lane);
        __syncthreads();  // everyone has read U[0:64,64:128] from S and the level-1 V blocks
        blk_store(VBLK(r, c), x, lane, 1.0f);
    }
    __syncthreads();
    {
        const int r = wv >> 1, c = 2 + (wv & 1);
        pf32x16 y = blk_mm<false>(VBLK(r, 2), VBLK(2, c), zero, lane);
        if (c == 3) y = blk_mm<false>(VBLK(r, 3), VBLK(3, 3), y, lane);
        blk_store(PBLK(S, r, c), y, lane, -1.0f);
    }
    __syncthreads();
    LLMC_STAMP(11);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e4 = tid + 256 * q;
        const int i = e4 >> 5, j = (e4 & 31) * 4;
        const int bi = i >> 5, bj = j >> 5;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bi == bj) v4 = *reinterpret_cast<const float4*>(Vd + (i & 31) * PLD + j);             // zeros below diag
        else if (bi < 2 && bj >= 2) v4 = *reinterpret_cast<const float4*>(S + i * PLD + j);       // level-2 result
        else if (bi < bj) v4 = *reinterpret_cast<const float4*>(S + (bj * 32 + (i & 31)) * PLD + bi * 32 + (j & 31));
        *reinterpret_cast<float4*>(Vout + i * NB + j) = v4;
    }
    LLMC_STAMP(12);
}
#undef PBLK

// gemm6 skips the zero part of a triangular operand at 256 granularity: the 128 x 128 block below the diagonal inside
// every 256-aligned diagonal tile must really be zero (nothing else reads that part of the work matrix)
__global__ __launch_bounds__(256) void k_zero_subdiag(float* __restrict__ W, int64_t ld, int K) {
    const int r0 = blockIdx.x * 256 + 128, c0 = blockIdx.x * 256;
    for (int e = threadIdx.x; e < 128 * 128; e += 256) {
        const int i = r0 + (e >> 7), j = c0 + (e & 127);
        if (i < K && j < K) W[(int64_t)i * ld + j] = 0.0f;
    }
}

// copy the inverted diagonal blocks into the work matrix (upper), before the doubling levels
__global__ __launch_bounds__(256) void k_place_diag_inv(float* __restrict__ W, int64_t ld, int K,
                                                        const float* __restrict__ Vbuf, int b0) {
    const int b = b0 + blockIdx.x;
    const int k0 = b * NB;
    const int nb = min(NB, K - k0);
    const float* V = Vbuf + (int64_t)b * NB * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 7, j = e & 127;
        if (i < nb && j < nb) W[(int64_t)(k0 + i) * ld + k0 + j] = V[e];
    }
}

// ---------------------------------------------------------------------------------------------
// K2 helpers
// ---------------------------------------------------------------------------------------------
// pass 1: dead fix on the diagonal, dead flags, sum of the fixed diagonal (fixed-order tree: deterministic)
__global__ __launch_bounds__(1024) void k_diag_fix(float* __restrict__ H, int K, uint8_t* __restrict__ dead,
                                                   float* __restrict__ diag_mean) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < K; i += 1024) {
        float d = H[(int64_t)i * K + i];
        uint8_t dd = (d == 0.0f);
        if (dd) {
            d = 1.0f;
            H[(int64_t)i * K + i] = 1.0f;
        }
        dead[i] = dd;
        s += (double)d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *diag_mean = (float)(red[0] / (double)K);
}

// Hout[i][j] = H[perm[i]][perm[j]] + (i == j) * percdamp * mean(diag)
__global__ __launch_bounds__(256) void k_gather_h(const float* __restrict__ H, int K,
                                                  const int64_t* __restrict__ perm, float percdamp,
                                                  const float* __restrict__ diag_mean, float* __restrict__ Hout) {
    const int i = blockIdx.y;
    const int64_t pi = perm ? perm[i] : i;
    const float damp = percdamp * (*diag_mean);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < K; j += gridDim.x * 256) {
        const int64_t pj = perm ? perm[j] : j;
        float v = H[pi * K + pj];
        if (i == j) v += damp;
        Hout[(int64_t)i * K + j] = v;
    }
}

// Wout[r][j] = dead[perm[j]] ? 0 : float(W[r][perm[j]])
template <typename T>
__global__ __launch_bounds__(256) void k_gather_w(const T* __restrict__ W, int64_t R, int K,
                                                  const int64_t* __restrict__ perm,
                                                  const uint8_t* __restrict__ dead, float* __restrict__ Wout) {
    const int64_t r = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < K; j += gridDim.x * 256) {
        const int64_t pj = perm ? perm[j] : j;
        float v = dead[pj] ? 0.0f : to_f32<T>(W[r * K + pj]);
        Wout[r * K + j] = v;
    }
}


// LDS-staged gathers (K % 4 == 0, K*4 <= 160 KB, i.e. K <= 40960: a 70B down_proj row, K = 28672, is 112 KB): a workgroup
// reads ONE source row contiguously (16-B loads), keeps it in LDS as fp32, and writes the permuted row contiguously (16-B
// stores) — both HBM streams coalesced; the random access happens in LDS. The element-wise kernels above remain for other
// shapes.
template <typename T> __device__ __forceinline__ float4 load4_f32(const T* p);
template <> __device__ __forceinline__ float4 load4_f32<float>(const float* p) {
    return *reinterpret_cast<const float4*>(p);
}
template <> __device__ __forceinline__ float4 load4_f32<f16_t>(const f16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f16_t h[4];
    __builtin_memcpy(h, &u, 8);
    return make_float4(to_f32<f16_t>(h[0]), to_f32<f16_t>(h[1]), to_f32<f16_t>(h[2]), to_f32<f16_t>(h[3]));
}
template <> __device__ __forceinline__ float4 load4_f32<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    bf16_t h[4];
    __builtin_memcpy(h, &u, 8);
    return make_float4(to_f32<bf16_t>(h[0]), to_f32<bf16_t>(h[1]), to_f32<bf16_t>(h[2]), to_f32<bf16_t>(h[3]));
}

// MODE 0: out[r][j] = in[r][idx[j]]                                   (column gather)
// MODE 1: out[r][j] = dead[idx[j]] ? 0 : in[r][idx[j]]                (k_gather_w)
// MODE 2: out[r][j] = in[idx[r]][idx[j]] + (r == j) * damp            (k_gather_h)
// A workgroup handles GATHER_ROWS consecutive output rows and keeps the column indices of its threads in registers across
// them (round 4: with one row per workgroup every row re-read the whole int64 index vector, twice the bytes of the row itself).
static constexpr int GATHER_ROWS = 8;
static constexpr int GATHER_MAXJ = 7;     // 4 * 512 * 7 = 14336 columns per register pass (28 index registers); wider rows take more passes
template <typename T, int MODE>
__global__ __launch_bounds__(512) void k_gather_lds(const T* __restrict__ in, int K, const int64_t* __restrict__ idx,
                                                   const uint8_t* __restrict__ dead, float percdamp,
                                                   const float* __restrict__ diag_mean, float* __restrict__ out, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) float grow[];
    const float damp = MODE == 2 ? percdamp * (*diag_mean) : 0.0f;
    const int64_t r_begin = (int64_t)blockIdx.x * GATHER_ROWS;
    const int64_t r_end = r_begin + GATHER_ROWS < R ? r_begin + GATHER_ROWS : R;
    for (int c_pass = 0; c_pass < K; c_pass += 4 * 512 * GATHER_MAXJ) {
        // this thread's columns of the pass: c = c_pass + 4 * tid + 2048 * j
        int pj[GATHER_MAXJ][4];
        uint32_t deadbits = 0;
#pragma unroll
        for (int j = 0; j < GATHER_MAXJ; ++j) {
            const int c = c_pass + 4 * (int)threadIdx.x + 4 * 512 * j;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int v = 0;
                if (c < K) {
                    v = idx ? (int)idx[c + t] : c + t;
                    if (MODE == 1 && dead[v]) deadbits |= 1u << (4 * j + t);
                }
                pj[j][t] = v;
            }
        }
        for (int64_t r = r_begin; r < r_end; ++r) {
            const int64_t sr = (MODE == 2 && idx) ? idx[r] : r;
            const T* src = in + sr * K;
            __syncthreads();          // the previous row's gathers are done with the LDS row
            for (int c = 4 * threadIdx.x; c < K; c += 4 * 512) *reinterpret_cast<float4*>(grow + c) = load4_f32<T>(src + c);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < GATHER_MAXJ; ++j) {
                const int c = c_pass + 4 * (int)threadIdx.x + 4 * 512 * j;
                if (c >= K) break;
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float x = grow[pj[j][t]];
                    if (MODE == 1 && ((deadbits >> (4 * j + t)) & 1u)) x = 0.0f;
                    if (MODE == 2 && r == c + t) x += damp;
                    v[t] = x;
                }
                *reinterpret_cast<float4*>(out + r * K + c) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

static constexpr int GATHER_LDS_MAX = 160 * 1024;   // the whole LDS of a CU (one workgroup per CU above 80 KB)

template <typename T, int MODE>
static int gather_lds_launch(const T* in, int64_t R, int64_t K, const int64_t* idx, const uint8_t* dead,
                             float percdamp, const float* diag_mean, float* out, hipStream_t st) {
    // rows above 64 KB need the kernel's dynamic-LDS ceiling raised once (per device and instantiation)
    if (K * 4 > 65536)
        if (int rc = ensure_dynamic_lds((const void*)k_gather_lds<T, MODE>, GATHER_LDS_MAX)) return rc;
    hipLaunchKernelGGL((k_gather_lds<T, MODE>), dim3((unsigned)ceil_div64(R, GATHER_ROWS)), dim3(512), (size_t)K * 4, st, in,
                       (int)K, idx, dead, percdamp, diag_mean, out, R);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
static inline bool gather_lds_ok(int64_t K, const void* in, const void* out, int esz) {
    return K % 4 == 0 && K * 4 <= GATHER_LDS_MAX && ((uintptr_t)in % (4 * esz)) == 0 && ((uintptr_t)out & 15) == 0;
}

}  // namespace llmc

using namespace llmc;

extern "C" int llmc_gather_cols(const float* in, int64_t R, int64_t K, const int64_t* idx, float* out,
                                llmc_stream_t stream) {
    LLMC_REQUIRE(in && idx && out && R > 0 && K > 0, "gather_cols: null/empty argument");
    LLMC_REQUIRE(in != out, "gather_cols: out must not alias in");
    hipStream_t st = (hipStream_t)stream;
    if (gather_lds_ok(K, in, out, 4))
        return gather_lds_launch<float, 0>(in, R, K, idx, nullptr, 0.0f, nullptr, out, st);
    set_last_error_msg("gather_cols: K must be a multiple of 4 with K <= 40960 and 16-byte aligned rows");
    return LLMC_ENOTSUP;
}

extern "C" size_t llmc_hessian_prep_ws_bytes(int64_t K) {
    if (K <= 0) return 0;
    // mean + dead flags, then (16-B aligned) the reversed permutation of llmc_hessian_prep_rev
    return (size_t)(((K + 63) / 64) * 64 + 64) + (size_t)K * 8;
}

namespace llmc {
__global__ __launch_bounds__(256) void k_reverse_perm(const int64_t* __restrict__ perm, int K, int64_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K) out[i] = perm ? perm[K - 1 - i] : (int64_t)(K - 1 - i);
}
}  // namespace llmc

static int hessian_prep_impl(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm, float percdamp,
                             float* Hout, float* Wout, void* ws, llmc_stream_t stream, bool rev);

extern "C" int llmc_hessian_prep(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm,
                                 float percdamp, float* Hout, float* Wout, void* ws, llmc_stream_t stream) {
    return hessian_prep_impl(H, W, wdt, R, K, perm, percdamp, Hout, Wout, ws, stream, false);
}

// The same preparation with Hout written index-REVERSED: Hout[i][j] = Hp[K-1-i][K-1-j] (Hp = the permuted, damped matrix
// llmc_hessian_prep writes). For a symmetric H that is Hp reflected across its anti-diagonal — exactly the matrix
// llmc_chol_inv_upper builds first from Hp with a transposing pass over K^2 floats — at no cost: it is the same gather with
// the permutation read backwards. llmc_chol_inv_upper_rev takes it. W is gathered with `perm` as always.
extern "C" int llmc_hessian_prep_rev(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm,
                                     float percdamp, float* Hout, float* Wout, void* ws, llmc_stream_t stream) {
    return hessian_prep_impl(H, W, wdt, R, K, perm, percdamp, Hout, Wout, ws, stream, true);
}

static int hessian_prep_impl(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm, float percdamp,
                             float* Hout, float* Wout, void* ws, llmc_stream_t stream, bool rev) {
    LLMC_REQUIRE(H && ws && K > 0, "hessian_prep: null/empty argument");
    LLMC_REQUIRE((W && Wout && R > 0) || (!W && !Wout), "hessian_prep: W and Wout go together");
    LLMC_REQUIRE(dtype_ok(wdt), "hessian_prep: bad dtype");
    LLMC_REQUIRE(Hout != H, "hessian_prep: Hout must not alias H");
    LLMC_REQUIRE(K < (1ll << 31) && R < 65536ll * 32768ll, "hessian_prep: shape too large");
    hipStream_t st = (hipStream_t)stream;
    float* diag_mean = (float*)ws;
    uint8_t* dead = (uint8_t*)ws + 64;
    hipLaunchKernelGGL(k_diag_fix, dim3(1), dim3(1024), 0, st, H, (int)K, dead, diag_mean);
    LLMC_LAUNCH_CHECK();
    int gx = (int)ceil_div64(K, 256 * 4);
    if (Hout) {
        const int64_t* hperm = perm;
        if (rev) {     // the permutation read backwards (identity: K-1 .. 0): Hout comes out reflected across the anti-diagonal
            int64_t* pr = (int64_t*)((char*)ws + (((K + 63) / 64) * 64 + 64));
            hipLaunchKernelGGL(k_reverse_perm, dim3((unsigned)ceil_div64(K, 256)), dim3(256), 0, st, perm, (int)K, pr);
            LLMC_LAUNCH_CHECK();
            hperm = pr;
        }
        if (gather_lds_ok(K, H, Hout, 4)) {
            int rc = gather_lds_launch<float, 2>(H, K, K, hperm, nullptr, percdamp, diag_mean, Hout, st);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(k_gather_h, dim3(gx, (unsigned)K), dim3(256), 0, st, (const float*)H, (int)K, hperm,
                               percdamp, (const float*)diag_mean, Hout);
            LLMC_LAUNCH_CHECK();
        }
    }
    if (W && gather_lds_ok(K, W, Wout, wdt == LLMC_F32 ? 4 : 2)) {
        int rc;
        if (wdt == LLMC_F16) rc = gather_lds_launch<f16_t, 1>((const f16_t*)W, R, K, perm, dead, 0.0f, nullptr, Wout, st);
        else if (wdt == LLMC_BF16)
            rc = gather_lds_launch<bf16_t, 1>((const bf16_t*)W, R, K, perm, dead, 0.0f, nullptr, Wout, st);
        else rc = gather_lds_launch<float, 1>((const float*)W, R, K, perm, dead, 0.0f, nullptr, Wout, st);
        return rc;
    }
    // grid.y is limited to 65535: loop over row slabs
    for (int64_t r0 = 0; W && r0 < R; r0 += 32768) {
        int64_t rows = R - r0 < 32768 ? R - r0 : 32768;
        dim3 grid(gx, (unsigned)rows);
        if (wdt == LLMC_F16)
            hipLaunchKernelGGL((k_gather_w<f16_t>), grid, dim3(256), 0, st, (const f16_t*)W + r0 * K, rows, (int)K,
                               perm, (const uint8_t*)dead, Wout + r0 * K);
        else if (wdt == LLMC_BF16)
            hipLaunchKernelGGL((k_gather_w<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)W + r0 * K, rows, (int)K,
                               perm, (const uint8_t*)dead, Wout + r0 * K);
        else
            hipLaunchKernelGGL((k_gather_w<float>), grid, dim3(256), 0, st, (const float*)W + r0 * K, rows, (int)K,
                               perm, (const uint8_t*)dead, Wout + r0 * K);
        LLMC_LAUNCH_CHECK();
    }
    return LLMC_OK;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// doubling levels of the triangular inverse from this block size on run on the one-wave-per-SIMD GEMM (gemm6_launch)
static constexpr int64_t GEMM6_MIN_H = 4096;
static size_t gemm6_bytes(int64_t K) {
    size_t mx = 0;
    for (int64_t h = NB; h < K; h *= 2)
        if (h >= GEMM6_MIN_H) {
            const size_t b = gemm6_ws_bytes((int)h, (int)h, (int)h);
            if (b > mx) mx = b;
        }
    return align256(mx);
}

extern "C" size_t llmc_chol_inv_upper_ws_bytes(int64_t K) {
    if (K <= 0) return 0;
    size_t work = align256((size_t)K * K * 4);
    size_t vbuf = align256((size_t)ceil_div64(K, NB) * NB * NB * 4);
    size_t xbuf = align256((size_t)(K / 2 + NB) * (K / 2 + NB) * 4);
    return work + vbuf + xbuf + gemm6_bytes(K);
}

static int chol_inv_upper_impl(float* A, float* Uout, int64_t K64, void* ws, int32_t* info_dev, llmc_stream_t stream);

extern "C" int llmc_chol_inv_upper(float* A, int64_t K64, void* ws, int32_t* info_dev, llmc_stream_t stream) {
    return chol_inv_upper_impl(A, nullptr, K64, ws, info_dev, stream);
}

// The same factor from the index-reversed matrix llmc_hessian_prep_rev writes: Arev is factored IN PLACE (destroyed) and U is
// written to Uout (must not alias Arev): no transposing pass in front of the factorisation. Same workspace size as
// llmc_chol_inv_upper; same kernels in the same order on the same values: U is bit-identical to llmc_chol_inv_upper's on the
// un-reversed matrix whenever that matrix is exactly symmetric (every Hessian of this library is).
extern "C" int llmc_chol_inv_upper_rev(float* Arev, float* Uout, int64_t K64, void* ws, int32_t* info_dev, llmc_stream_t stream) {
    LLMC_REQUIRE(Uout && Uout != Arev && ((uintptr_t)Uout & 15) == 0, "chol_inv_upper_rev: Uout must be a separate 16-B aligned buffer");
    return chol_inv_upper_impl(Arev, Uout, K64, ws, info_dev, stream);
}

static int chol_inv_upper_impl(float* A, float* Uout, int64_t K64, void* ws, int32_t* info_dev, llmc_stream_t stream) {
    LLMC_REQUIRE(A && ws && info_dev && K64 > 0, "chol_inv_upper: null/empty argument");
    LLMC_REQUIRE(K64 % 4 == 0 && K64 < (1 << 30), "chol_inv_upper: K must be a multiple of 4");
    LLMC_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)ws & 255) == 0, "chol_inv_upper: alignment");
    hipStream_t st = (hipStream_t)stream;
    const int K = (int)K64;
    const bool rev = Uout != nullptr;          // A already holds the reflected matrix: work in it
    float* Wk = rev ? A : (float*)ws;
    float* Vbuf = (float*)((char*)ws + align256((size_t)K * K * 4));
    float* Xbuf = (float*)((char*)Vbuf + align256((size_t)ceil_div64(K, NB) * NB * NB * 4));
    void* G6buf = (char*)Xbuf + align256((size_t)(K / 2 + NB) * (K / 2 + NB) * 4);
    const bool use_g6 = !opt(OPT_K3_NO_GEMM6);
    LLMC_HIP_CHECK(hipMemsetAsync(info_dev, 0, 4, st));
    if (int rc = ensure_dynamic_lds((const void*)k_potrf_inv, (NB * PLD + 32 * PLD + 64) * (int)sizeof(float))) return rc;

    dim3 tgrid((K + 31) / 32, (K + 31) / 32);
    if (!rev) {
        hipLaunchKernelGGL(k_antitranspose, tgrid, dim3(256), 0, st, (const float*)A, Wk, K, 0);
        LLMC_LAUNCH_CHECK();
    }

    const int nblk = (K + NB - 1) / NB;
    // ---- blocked upper Cholesky Wk = U'^T U', two-level: 128-wide factor steps inside 512-wide outer blocks.
    // Inside an outer block every step updates only the rows of that block (Kd = 128, few rows); the rows
    // beyond it receive ONE symmetric update with Kd = 512 per outer block, which is where the flops are and
    // runs the fp32-MFMA GEMM at its long-K efficiency instead of its short-K one (tools/bench_sgemm.py).
    const int NBO = 4 * NB;
    SideStream* side = (!helper_streams_enabled()) ? nullptr : side_stream_for(st);
    bool pending_side = false;
    // The large products of K3 (far updates, triangular-inverse levels >= 512) run as split-bf16 products on the 16-bit
    // MFMA pipe (gemm3.hip: fp32-level accuracy, 1.3-1.5x the fp32-MFMA kernel). option k3_fp32 keeps everything on
    // the fp32 MFMA path.
    const bool k3_x3 = !opt(OPT_K3_FP32);
    const bool use_x3u = k3_x3, use_x3 = k3_x3, use_x3t = k3_x3;
    const bool use_planes = !opt(OPT_K3_NO_PLANES);
    const bool merge_far = !opt(OPT_K3_SPLIT_FAR);
    const size_t xbuf_bytes = (size_t)(K / 2 + NB) * (K / 2 + NB) * 4;
    // Inside an outer block the columns split into NEAR (the block's own, which the next factor step needs) and FAR (all
    // the columns to its right, needed by the later far panel solves and by the block's far update). The near panel solve
    // and near update stay on the caller's stream between the diagonal factorisations — three small latency-bound
    // kernels per step; the far panel solve and far in-block update of the same step (128 x nfar products: the bulk of a
    // step's work) run on the helper stream behind an event, concurrently with the next steps' chain. Same kernels, same
    // arithmetic per element and the same order of the updates an element receives (all of step c before step c + 1 on
    // either stream), so the factor is bit-identical with or without the helper stream.
    auto panel_solve = [&](const float* Vb, int c0, int nb, int col0, int ncols, hipStream_t s_) -> int {
        if (ncols <= 0) return LLMC_OK;
        float* P = Wk + (size_t)c0 * K + col0;     // rows c0..c0+nb, cols col0..col0+ncols
        SgemmArgs g{};
        // P = V^T P  (op(A)[i][k] = V[k][i], lower triangular)
        g.A = Vb; g.lda = NB; g.B = P; g.ldb = K; g.C = P; g.ldc = K;
        g.M = g.M_last = nb; g.N = g.N_last = ncols; g.Kd = g.Kd_last = nb;
        g.epilogue = SG_SET; g.a_lower = 1; g.batch = 1;
        return sgemm_launch(g, true, false, s_);
    };
    // rows c0+nb .. oend of the trailing matrix, columns col0 .. col0+ncols:  C -= P[:, rows]^T P[:, cols]
    auto inblock_update = [&](int c0, int nb, int oend, int col0, int ncols, hipStream_t s_) -> int {
        const int mrows = oend - (c0 + nb);
        if (mrows <= 0 || ncols <= 0) return LLMC_OK;
        SgemmArgs u{};
        u.A = Wk + (size_t)c0 * K + c0 + nb; u.lda = K;
        u.B = Wk + (size_t)c0 * K + col0; u.ldb = K;
        u.C = Wk + (size_t)(c0 + nb) * K + col0; u.ldc = K;
        u.M = u.M_last = mrows; u.N = u.N_last = ncols; u.Kd = u.Kd_last = nb;
        u.epilogue = SG_SUB; u.c_upper_only = col0 == c0 + nb ? 1 : 0; u.batch = 1;   // the far columns lie right of every row
        return sgemm_launch(u, true, false, s_);
    };
    for (int k0 = 0; k0 < K; k0 += NBO) {
        const int nbo = K - k0 < NBO ? K - k0 : NBO;
        const int oend = k0 + nbo;
        const int nfar = K - oend;
        for (int c0 = k0; c0 < oend; c0 += NB) {
            const int b = c0 / NB;
            const int nb = K - c0 < NB ? K - c0 : NB;
            float* Vb = Vbuf + (size_t)b * NB * NB;
            hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), (NB * PLD + 32 * PLD + 64) * sizeof(float), st, Wk, (int64_t)K, c0,
                               nb, Vb, info_dev);
            LLMC_LAUNCH_CHECK();
            if (K - c0 - nb <= 0) break;
            const int nnear = oend - (c0 + nb);
            if (!side || nfar <= 0) {
                // one stream: near and far columns in ONE panel solve and ONE update per step (the split costs two more
                // launches per step, 1.3 ms over a K = 14336 factorisation, and buys nothing without a second stream)
                int rc = panel_solve(Vb, c0, nb, c0 + nb, K - c0 - nb, st);
                if (rc) return rc;
                rc = inblock_update(c0, nb, oend, c0 + nb, K - c0 - nb, st);
                if (rc) return rc;
                continue;
            }
            int rc = panel_solve(Vb, c0, nb, c0 + nb, nnear, st);
            if (rc) return rc;
            rc = fork_to_side(side, st);      // the far part of this step: behind the near panel, beside the rest of the chain
            if (rc) return rc;
            pending_side = true;
            rc = panel_solve(Vb, c0, nb, oend, nfar, side->side);
            if (rc) return rc;
            rc = inblock_update(c0, nb, oend, oend, nfar, side->side);
            if (rc) return rc;
            rc = inblock_update(c0, nb, oend, c0 + nb, nnear, st);
            if (rc) return rc;
        }
        if (nfar > 0) {
            // far trailing update T -= P^T P with P = rows k0..oend, cols oend..K (Kd = nbo), in two parts: the rows
            // of the NEXT outer block on the main stream (its factor steps need them), the rows below on the side
            // stream, overlapped with the next outer block's latency-bound diagonal / panel kernels.
            float* P = Wk + (size_t)k0 * K + oend;
            const int m1 = nfar < NBO ? nfar : NBO;
            if (side && pending_side) {          // the block's far panels (and the previous block's side update) are complete
                int rc = join_from_side(side, st);
                if (rc) return rc;
                pending_side = false;
            }
            SgemmArgs u{};
            u.A = P; u.lda = K; u.B = P; u.ldb = K;
            u.C = Wk + (size_t)oend * K + oend; u.ldc = K;
            u.M = u.M_last = m1; u.N = u.N_last = nfar; u.Kd = u.Kd_last = nbo;
            u.epilogue = SG_SUB; u.c_upper_only = 1; u.batch = 1;
            const int m2 = nfar - m1;
            SgemmArgs v{};
            v.A = P + m1; v.lda = K; v.B = P + m1; v.ldb = K;
            v.C = Wk + (size_t)(oend + m1) * K + oend + m1; v.ldc = K;
            v.M = v.M_last = m2; v.N = v.N_last = m2; v.Kd = v.Kd_last = nbo;
            v.epilogue = SG_SUB; v.c_upper_only = 1; v.batch = 1;
            // Large far updates: the panel P is split ONCE into its three bf16 planes (in Xbuf, which only the inverse
            // levels use) and the products copy planes instead of splitting P again in every tile (gemm3.hip, k_gemm3s).
            // Same planes, same MFMA order: the factor is bit-identical either way. The previous block's side update
            // (which reads the previous planes) was joined above.
            // Without a helper stream the two parts are ONE product: the upper triangle of the whole trailing matrix (u's tiles
            // are the first two tile rows of it, v's start at row and column 512 = whole tiles): one launch less per outer block
            // and u's 216 tiles no longer run as a round of their own. Same tiles, same arithmetic.
            const bool merged = !side && use_x3 && use_x3u && m2 > 0 && merge_far;
            if (merged) u.M = u.M_last = nfar;
            if (use_x3 && use_planes && nfar % 8 == 0 && (size_t)3 * nbo * nfar * 2 <= xbuf_bytes) {
                SgemmArgs w = merged ? u : (m2 > 0 ? v : u);
                w.planesA = w.planesB = Xbuf; w.ldp = nfar; w.plane_stride = (int64_t)nbo * nfar;
                if (gemm3_uses_planes(w)) {
                    int rc = gemm3_split_planes(P, K, nbo, nfar, Xbuf, nfar, (int64_t)nbo * nfar, st);
                    if (rc) return rc;
                    u.planesA = u.planesB = Xbuf;
                    v.planesA = v.planesB = (const uint16_t*)Xbuf + m1;
                    u.ldp = v.ldp = nfar;
                    u.plane_stride = v.plane_stride = (int64_t)nbo * nfar;
                }
            }
            int rc = use_x3u ? gemm3_tn_launch(u, st) : sgemm_launch(u, true, false, st);
            if (rc) return rc;
            if (m2 > 0 && !merged) {
                if (side) {
                    rc = fork_to_side(side, st);   // P is final on main at this point
                    if (rc) return rc;
                    rc = use_x3 ? gemm3_tn_launch(v, side->side) : sgemm_launch(v, true, false, side->side);
                    if (rc) return rc;
                    pending_side = true;
                } else {
                    rc = use_x3 ? gemm3_tn_launch(v, st) : sgemm_launch(v, true, false, st);
                    if (rc) return rc;
                }
            }
        }
    }
    if (side && pending_side) {
        int rc = join_from_side(side, st);
        if (rc) return rc;
    }
    (void)nblk;
    // ---- V = U'^-1: inverted diagonal blocks, then doubling levels
    hipLaunchKernelGGL(k_place_diag_inv, dim3((K + NB - 1) / NB), dim3(256), 0, st, Wk, (int64_t)K, K, (const float*)Vbuf, 0);
    LLMC_LAUNCH_CHECK();
    if (use_x3t && use_g6 && K > GEMM6_MIN_H) {
        hipLaunchKernelGGL(k_zero_subdiag, dim3((K + 255) / 256), dim3(256), 0, st, Wk, (int64_t)K, K);
        LLMC_LAUNCH_CHECK();
    }
    for (int64_t h = NB; h < K; h *= 2) {
        const int npairs = (int)((K - h + 2 * h - 1) / (2 * h));  // pairs with a non-empty right block
        if (npairs <= 0) break;
        const int64_t o_last = (int64_t)(npairs - 1) * 2 * h;
        const int n2_last = (int)((K - o_last - h) < h ? (K - o_last - h) : h);
        const int64_t stride = 2 * h * ((int64_t)K + 1);
        // X = A^-1 C
        SgemmArgs x{};
        x.A = Wk; x.lda = K; x.sA = stride;                 // A^-1 at (o, o), upper
        x.B = Wk + h; x.ldb = K; x.sB = stride;             // C at (o, o+h)
        // X is [h x n2]: with one pair its leading dimension shrinks to n2 (keeps X within K^2/4 floats)
        const int64_t ldX = npairs == 1 ? ((n2_last + 3) / 4) * 4 : h;
        x.C = Xbuf; x.ldc = ldX; x.sC = h * h;
        x.M = x.M_last = (int)h; x.N = (int)h; x.N_last = n2_last; x.Kd = x.Kd_last = (int)h;
        x.epilogue = SG_SET; x.a_upper = 1; x.batch = npairs;
        const bool lvl_x3 = use_x3t && h >= 512;   // small levels are latency-bound: the fp32 kernels stay
        // large, deep levels: operands split once into stacked bf16 planes, product on the one-wave-per-SIMD GEMM
        const bool lvl_g6 = lvl_x3 && use_g6 && h >= GEMM6_MIN_H && h % 256 == 0 && n2_last % 256 == 0;
        int rc = LLMC_OK;
        if (lvl_g6) {
            for (int z = 0; z < npairs && !rc; ++z) {
                const int n2 = z == npairs - 1 ? n2_last : (int)h;
                rc = gemm6_launch(x.A + z * stride, K, x.B + z * stride, K, Xbuf + (int64_t)z * h * h, ldX, (int)h, n2, (int)h,
                                  1, 0, 1.0f, G6buf, st);
                if (rc) return rc;
                rc = gemm6_launch(Xbuf + (int64_t)z * h * h, ldX, Wk + h * ((int64_t)K + 1) + z * stride, K,
                                  Wk + h + z * stride, K, (int)h, n2, n2, 0, 1, -1.0f, G6buf, st);
            }
            if (rc) return rc;
            continue;
        }
        rc = lvl_x3 ? gemm3_launch(x, false, st) : sgemm_launch(x, false, false, st);
        if (rc) return rc;
        // C = -X B^-1
        SgemmArgs y{};
        y.A = Xbuf; y.lda = ldX; y.sA = h * h;
        y.B = Wk + h * ((int64_t)K + 1); y.ldb = K; y.sB = stride;   // B^-1 at (o+h, o+h), upper
        y.C = Wk + h; y.ldc = K; y.sC = stride;
        y.M = y.M_last = (int)h; y.N = (int)h; y.N_last = n2_last; y.Kd = (int)h; y.Kd_last = n2_last;
        y.epilogue = SG_NEG; y.b_upper = 1; y.batch = npairs;
        rc = lvl_x3 ? gemm3_launch(y, false, st) : sgemm_launch(y, false, false, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_antitranspose, tgrid, dim3(256), 0, st, (const float*)Wk, rev ? Uout : A, K, 1);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

