// sgemm.hip — fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain). See sgemm.h.
// Workgroup tile 128x128, 4 waves as 2x2 (64x64 each = 2x2 MFMA accumulators), K-step 16, operands
// staged k-major in LDS ([k][row]) so that an MFMA operand read (32 consecutive rows at one k) is a
// conflict-free ds_read_b32; a memory layout that is row-major along k is transposed by the staging
// writes. Register-prefetched double buffering, one barrier per K-step.
#include <stdlib.h>
#include "sgemm.h"

namespace llmc {

static constexpr int GB = 128;   // tile edge
static constexpr int GK = 16;    // K-step
static constexpr int GLD = 132;  // LDS row stride (floats): 16-B aligned rows, 2-way at most on transposing writes

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Stage {
    float4 v[2];
};

// memory is k-major: src[k*ld + c]; tile rows k0..k0+15, cols c0..c0+127. EDGE = false: the tile is known to be
// fully inside the operand (no bounds checks, no control flow around the loads).
template <bool EDGE>
__device__ __forceinline__ Stage load_kmajor(const float* src, int64_t ld, int k0, int c0, int kmax, int cmax,
                                             int tid) {
    Stage s;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int idx = tid + 256 * h;
        int k = k0 + (idx >> 5);
        int c = c0 + 4 * (idx & 31);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!EDGE) {
            v = *reinterpret_cast<const float4*>(src + (int64_t)k * ld + c);
        } else if (k < kmax) {
            const float* p = src + (int64_t)k * ld + c;
            if (c + 3 < cmax) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (c < cmax) v.x = p[0];
                if (c + 1 < cmax) v.y = p[1];
                if (c + 2 < cmax) v.z = p[2];
            }
        }
        s.v[h] = v;
    }
    return s;
}
__device__ __forceinline__ void store_kmajor(float* lds, const Stage& s, int tid) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int idx = tid + 256 * h;
        *reinterpret_cast<float4*>(lds + (idx >> 5) * GLD + 4 * (idx & 31)) = s.v[h];
    }
}
// memory is c-major: src[c*ld + k]; transposed by the LDS writes
template <bool EDGE>
__device__ __forceinline__ Stage load_cmajor(const float* src, int64_t ld, int k0, int c0, int kmax, int cmax,
                                             int tid) {
    Stage s;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int idx = tid + 256 * h;
        int c = c0 + (idx >> 2);
        int k = k0 + 4 * (idx & 3);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!EDGE) {
            v = *reinterpret_cast<const float4*>(src + (int64_t)c * ld + k);
        } else if (c < cmax) {
            const float* p = src + (int64_t)c * ld + k;
            if (k + 3 < kmax) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (k < kmax) v.x = p[0];
                if (k + 1 < kmax) v.y = p[1];
                if (k + 2 < kmax) v.z = p[2];
            }
        }
        s.v[h] = v;
    }
    return s;
}
__device__ __forceinline__ void store_cmajor(float* lds, const Stage& s, int tid) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int idx = tid + 256 * h;
        int c = idx >> 2, k = 4 * (idx & 3);
        lds[(k + 0) * GLD + c] = s.v[h].x;
        lds[(k + 1) * GLD + c] = s.v[h].y;
        lds[(k + 2) * GLD + c] = s.v[h].z;
        lds[(k + 3) * GLD + c] = s.v[h].w;
    }
}

template <bool TA, bool TB, bool PHASED, bool EDGE>
__global__ __launch_bounds__(256, 2) void k_sgemm(SgemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * GK * GLD];  // [buf][A|B][k][row]
    const int z = blockIdx.z;
    const float* A = a.A + (int64_t)z * a.sA;
    const float* B = a.B + (int64_t)z * a.sB;
    float* C = a.C + (int64_t)z * a.sC;
    const bool last = z == a.batch - 1;
    const int M = last ? a.M_last : a.M, N = last ? a.N_last : a.N, Kd = last ? a.Kd_last : a.Kd;
    // b_upper: a tile's K range grows with its column, so columns are dealt heaviest first (shorter tail)
    const int bx = a.b_upper ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int i0 = blockIdx.y * GB, j0 = bx * GB;
    if (i0 >= M || j0 >= N) return;
    if (a.c_upper_only && j0 + GB <= i0) return;  // tile strictly below the diagonal

    int kb = 0, ke = Kd;
    if (a.a_upper) kb = (i0 / GK) * GK;
    if (a.a_lower) ke = min(ke, i0 + GB);
    if (a.b_upper) ke = min(ke, j0 + GB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    // phased mode: the C tile lives in registers for the whole K loop
    f32x16 cv[2][2];
    const bool phased = PHASED && a.phase_len > 0 && a.epilogue == SG_SUB;
    if (PHASED && phased) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = j0 + wn * 64 + n * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    cv[m][n][r] = (row < M && col < N) ? C[(int64_t)row * a.ldc + col] : 0.0f;
                }
            }
    }

    auto loadA = [&](int k0) {
        // op(A)[i][k]: TA -> stored [Kd x M] k-major ; else stored [M x Kd] row(i)-major
        return TA ? load_kmajor<EDGE>(A, a.lda, k0, i0, ke, M, tid) : load_cmajor<EDGE>(A, a.lda, k0, i0, ke, M, tid);
    };
    auto loadB = [&](int k0) {
        // op(B)[k][j]: !TB -> stored [Kd x N] k-major ; TB -> stored [N x Kd]
        return TB ? load_cmajor<EDGE>(B, a.ldb, k0, j0, ke, N, tid) : load_kmajor<EDGE>(B, a.ldb, k0, j0, ke, N, tid);
    };
    auto storeA = [&](int buf, const Stage& s) {
        float* p = lds + buf * (2 * GK * GLD);
        if (TA) store_kmajor(p, s, tid); else store_cmajor(p, s, tid);
    };
    auto storeB = [&](int buf, const Stage& s) {
        float* p = lds + buf * (2 * GK * GLD) + GK * GLD;
        if (TB) store_cmajor(p, s, tid); else store_kmajor(p, s, tid);
    };

    if (kb < ke) {
        // register prefetch of depth two: the loads of K-step n+2 are issued before the MFMAs of step n, so a load
        // has two MFMA phases (~1.7 us) to land before it is written to LDS (one phase is less than the HBM latency
        // under load and stalled every step)
        Stage ra[2], rb[2];
        ra[0] = loadA(kb);
        rb[0] = loadB(kb);
        storeA(0, ra[0]);
        storeB(0, rb[0]);
        if (kb + GK < ke) {
            ra[1] = loadA(kb + GK);
            rb[1] = loadB(kb + GK);
        }
        __syncthreads();
        auto step = [&](int k0, int cur, Stage& fa_next2, Stage& fb_next2, const Stage& fa_next, const Stage& fb_next) {
            const bool more = k0 + GK < ke;
            if (k0 + 2 * GK < ke) {
                fa_next2 = loadA(k0 + 2 * GK);
                fb_next2 = loadB(k0 + 2 * GK);
            }
            const float* pa = lds + cur * (2 * GK * GLD) + (lane >> 5) * GLD + wm * 64 + (lane & 31);
            const float* pb = lds + cur * (2 * GK * GLD) + GK * GLD + (lane >> 5) * GLD + wn * 64 + (lane & 31);
            // operand fragments double-buffered in registers: the LDS reads of pair kk+1 are in flight while the four
            // MFMAs of pair kk run (reusing one register set exposes the LDS latency before every group of MFMAs)
            float fa[2][2], fb[2][2];
            fa[0][0] = pa[0];
            fa[0][1] = pa[32];
            fb[0][0] = pb[0];
            fb[0][1] = pb[32];
#pragma unroll
            for (int kk = 0; kk < GK / 2; ++kk) {
                const int c = kk & 1;
                if (kk + 1 < GK / 2) {
                    fa[c ^ 1][0] = pa[2 * (kk + 1) * GLD];
                    fa[c ^ 1][1] = pa[2 * (kk + 1) * GLD + 32];
                    fb[c ^ 1][0] = pb[2 * (kk + 1) * GLD];
                    fb[c ^ 1][1] = pb[2 * (kk + 1) * GLD + 32];
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][m], fb[c][n], acc[m][n], 0, 0, 0);
            }
            // machine schedule: reads of pairs 0 and 1, then [4 MFMAs of pair kk, reads of pair kk+2] ... (left alone the
            // scheduler sinks every read to just before its MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int kk = 0; kk < GK / 2; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (kk + 2 < GK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            if (PHASED && phased && ((k0 + GK - kb) % a.phase_len == 0 || !more)) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            cv[m][n][r] = cv[m][n][r] - acc[m][n][r];
                            acc[m][n][r] = 0.0f;
                        }
            }
            if (more) {
                storeA(cur ^ 1, fa_next);
                storeB(cur ^ 1, fb_next);
            }
            __syncthreads();
        };
        for (int k0 = kb; k0 < ke; k0 += 2 * GK) {
            step(k0, 0, ra[0], rb[0], ra[1], rb[1]);                       // stage in buffer 0; next stage sits in r[1]
            if (k0 + GK < ke) step(k0 + GK, 1, ra[1], rb[1], ra[0], rb[0]);
        }
    }

    // epilogue: acc[m][n][r] -> row i0 + wm*64 + m*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), col j0 + wn*64 + n*32 + (lane&31)
    // SG_SUB without the phased path: all C loads of a 32x32 block are issued before its first store (a fused
    // load-subtract-store loop is ordered load, store, load, ... by possible aliasing: 64 serial round trips).
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = j0 + wn * 64 + n * 32 + (lane & 31);
            const bool colok = col < N;
            float old[16];
            if (!(PHASED && phased) && a.epilogue == SG_SUB) {
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
                if (PHASED && phased) *pc = cv[m][n][r];
                else if (a.epilogue == SG_SUB) *pc = old[r] - v;
                else if (a.epilogue == SG_SET) *pc = v;
                else *pc = -v;
            }
        }
}


// ---------------------------------------------------------------------------------------------------------
// Short-K variant (Kd <= 128, op(B) = N): the panel solves / in-block updates of the factorisation and the
// near updates of the GPTQ column loop are single 128-deep products whose cost is latency, not flops. Here a
// workgroup owns a 64x64 tile (4x the workgroups of the 128x128 kernel), fetches its WHOLE A and B panels with
// every load in flight at once (64 KB of LDS, one barrier), and runs one 64-step MFMA chain per wave. Same
// arithmetic as k_sgemm: one accumulator per element, ascending k from +0, then the epilogue's single rounding.
static constexpr int SB = 64;     // tile edge
static constexpr int SKD = 128;   // max K
static constexpr int SLB = 68;    // LDS row stride of k-major operands written as float4
static constexpr int SLT = 65;    // LDS row stride of the transposed (row-major in memory) A operand

template <bool TA>
__global__ __launch_bounds__(256) void k_sgemm_shortk(SgemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[SKD * SLB];
    __shared__ __attribute__((aligned(16))) float Bs[SKD * SLB];
    const int M = a.M, N = a.N, Kd = a.Kd;
    const int j0 = blockIdx.x * SB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    constexpr int LA = TA ? SLB : SLT;
    const int col = j0 + wn * 32 + (lane & 31);

    // ---- B panel once per workgroup: 8 float4 per thread, all issued before the first LDS write
    float4 vb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + 256 * q;
        const int k = idx >> 4, j = j0 + 4 * (idx & 15);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < Kd) {
            const float* p = a.B + (int64_t)k * a.ldb + j;
            if (j + 3 < N) v = *reinterpret_cast<const float4*>(p);
            else {
                if (j < N) v.x = p[0];
                if (j + 1 < N) v.y = p[1];
                if (j + 2 < N) v.z = p[2];
            }
        }
        vb[q] = v;
    }
    // ---- row tiles of this column block. gridDim.y covers all of them (one iteration each) except for an in-place
    // product (C aliases B, the panel solve P = V^T P): there ONE workgroup walks the row tiles, so that every row
    // of B it needs is in LDS before any row of C is written (separate workgroups would race: the tile of rows
    // 64..127 reads rows 0..63, which the tile of rows 0..63 overwrites).
    bool first = true;
    for (int i0 = blockIdx.y * SB; i0 < M; i0 += gridDim.y * SB) {
        if (a.c_upper_only && j0 + SB <= i0) continue;   // tile strictly below the diagonal (uniform)
        int ke = Kd;
        if (a.a_lower) ke = min(ke, i0 + SB);
        // C tile first (SG_SUB): its latency overlaps the operand loads
        float cv[16];
        if (a.epilogue == SG_SUB) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                cv[r] = (row < M && col < N) ? a.C[(int64_t)row * a.ldc + col] : 0.0f;
            }
        }
        float4 va[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + 256 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TA) {   // memory [Kd x M]: k = idx >> 4, 4 consecutive rows i
                const int k = idx >> 4, i = i0 + 4 * (idx & 15);
                if (k < ke) {
                    const float* p = a.A + (int64_t)k * a.lda + i;
                    if (i + 3 < M) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (i < M) v.x = p[0];
                        if (i + 1 < M) v.y = p[1];
                        if (i + 2 < M) v.z = p[2];
                    }
                }
            } else {    // memory [M x Kd]: row i = idx >> 5, 4 consecutive k
                const int i = i0 + (idx >> 5), k = 4 * (idx & 31);
                if (i < M) {
                    const float* p = a.A + (int64_t)i * a.lda + k;
                    if (k + 3 < ke) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (k < ke) v.x = p[0];
                        if (k + 1 < ke) v.y = p[1];
                        if (k + 2 < ke) v.z = p[2];
                    }
                }
            }
            va[q] = v;
        }
        if (!first) __syncthreads();   // the previous row tile's MFMAs have read As
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + 256 * q;
            if (TA) {
                *reinterpret_cast<float4*>(As + (idx >> 4) * SLB + 4 * (idx & 15)) = va[q];
            } else {
                const int i = idx >> 5, k = 4 * (idx & 31);
                As[(k + 0) * SLT + i] = va[q].x;
                As[(k + 1) * SLT + i] = va[q].y;
                As[(k + 2) * SLT + i] = va[q].z;
                As[(k + 3) * SLT + i] = va[q].w;
            }
            if (first) *reinterpret_cast<float4*>(Bs + (idx >> 4) * SLB + 4 * (idx & 15)) = vb[q];
        }
        first = false;
        __syncthreads();

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float* pa = As + (lane >> 5) * LA + wm * 32 + (lane & 31);
        const float* pb = Bs + (lane >> 5) * SLB + wn * 32 + (lane & 31);
        const int npair = (ke + 1) >> 1;   // rows >= ke of the A panel are zero-filled
#pragma unroll 8
        for (int kk = 0; kk < npair; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kk * LA], pb[2 * kk * SLB], acc, 0, 0, 0);

        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= M) continue;
                float* pc = a.C + (int64_t)row * a.ldc + col;
                if (a.epilogue == SG_SUB) *pc = cv[r] - acc[r];
                else if (a.epilogue == SG_SET) *pc = acc[r];
                else *pc = -acc[r];
            }
        }
    }
}


// Phased form of the short-K kernel for latency-critical products with few tiles (the part of K4's far update that the
// next column group waits for): Kd = n * 128, C -= A[:, p] B[p, :] phase by phase exactly like k_sgemm's phased mode
// (each phase: one accumulator per element from +0 in ascending k, then one rounding C - acc), but on 64x64 tiles,
// i.e. 4x the workgroups, each streaming its panels in 128-deep chunks with the next chunk's loads in flight under
// the current chunk's MFMAs.
template <bool TA>
__global__ __launch_bounds__(256) void k_sgemm_shortk_phased(SgemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[SKD * SLB];
    __shared__ __attribute__((aligned(16))) float Bs[SKD * SLB];
    const int M = a.M, N = a.N, Kd = a.Kd;
    const int i0 = blockIdx.y * SB, j0 = blockIdx.x * SB;
    if (a.c_upper_only && j0 + SB <= i0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    constexpr int LA = TA ? SLB : SLT;
    const int col = j0 + wn * 32 + (lane & 31);
    float cv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        cv[r] = (row < M && col < N) ? a.C[(int64_t)row * a.ldc + col] : 0.0f;
    }
    float4 va[8], vb[8];
    auto gload = [&](int kc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + 256 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TA) {
                const int k = kc + (idx >> 4), i = i0 + 4 * (idx & 15);
                if (k < Kd) {
                    const float* p = a.A + (int64_t)k * a.lda + i;
                    if (i + 3 < M) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (i < M) v.x = p[0];
                        if (i + 1 < M) v.y = p[1];
                        if (i + 2 < M) v.z = p[2];
                    }
                }
            } else {
                const int i = i0 + (idx >> 5), k = kc + 4 * (idx & 31);
                if (i < M && k + 3 < Kd) v = *reinterpret_cast<const float4*>(a.A + (int64_t)i * a.lda + k);
            }
            va[q] = v;
            const int kb = kc + (idx >> 4), j = j0 + 4 * (idx & 15);
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kb < Kd) {
                const float* p = a.B + (int64_t)kb * a.ldb + j;
                if (j + 3 < N) w = *reinterpret_cast<const float4*>(p);
                else {
                    if (j < N) w.x = p[0];
                    if (j + 1 < N) w.y = p[1];
                    if (j + 2 < N) w.z = p[2];
                }
            }
            vb[q] = w;
        }
    };
    gload(0);
    const float* pa = As + (lane >> 5) * LA + wm * 32 + (lane & 31);
    const float* pb = Bs + (lane >> 5) * SLB + wn * 32 + (lane & 31);
    for (int kc = 0; kc < Kd; kc += SKD) {
        if (kc) __syncthreads();   // the previous chunk's MFMAs have read the panels
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + 256 * q;
            if (TA) {
                *reinterpret_cast<float4*>(As + (idx >> 4) * SLB + 4 * (idx & 15)) = va[q];
            } else {
                const int i = idx >> 5, k = 4 * (idx & 31);
                As[(k + 0) * SLT + i] = va[q].x;
                As[(k + 1) * SLT + i] = va[q].y;
                As[(k + 2) * SLT + i] = va[q].z;
                As[(k + 3) * SLT + i] = va[q].w;
            }
            *reinterpret_cast<float4*>(Bs + (idx >> 4) * SLB + 4 * (idx & 15)) = vb[q];
        }
        __syncthreads();
        if (kc + SKD < Kd) gload(kc + SKD);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 8
        for (int kk = 0; kk < SKD / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kk * LA], pb[2 * kk * SLB], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[r] = cv[r] - acc[r];
    }
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) a.C[(int64_t)row * a.ldc + col] = cv[r];
        }
    }
}

int sgemm_launch(const SgemmArgs& a, bool TA, bool TB, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return LLMC_OK;
    LLMC_REQUIRE((a.lda % 4 == 0) && (a.ldb % 4 == 0) && (((uintptr_t)a.A & 15) == 0) &&
                     (((uintptr_t)a.B & 15) == 0) && (a.sA % 4 == 0) && (a.sB % 4 == 0),
                 "sgemm: operands must be 16-B aligned with ld % 4 == 0");
    if (!TB && a.batch == 1 && a.Kd <= SKD && !a.a_upper && !a.b_upper && (a.phase_len == 0 || a.phase_len >= a.Kd) &&
        !opt(OPT_NO_SHORTK)) {
        const bool in_place = (const void*)a.C == (const void*)a.B;
        dim3 sgrid((a.N + SB - 1) / SB, in_place ? 1 : (a.M + SB - 1) / SB, 1);
        if (TA) hipLaunchKernelGGL((k_sgemm_shortk<true>), sgrid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_sgemm_shortk<false>), sgrid, dim3(256), 0, st, a);
        LLMC_LAUNCH_CHECK();
        return LLMC_OK;
    }
    // few-tile phased products (phase = 128) on 64x64 tiles: the latency-critical slice of K4's far update
    if (!TB && a.batch == 1 && a.epilogue == SG_SUB && a.phase_len == SKD && a.Kd % SKD == 0 && a.Kd > SKD &&
        !a.a_upper && !a.a_lower && !a.b_upper && (TA || a.lda % 4 == 0) &&
        (int64_t)((a.M + SB - 1) / SB) * ((a.N + SB - 1) / SB) <= 1024 && !opt(OPT_NO_SHORTK)) {
        dim3 sgrid((a.N + SB - 1) / SB, (a.M + SB - 1) / SB, 1);
        if (TA) hipLaunchKernelGGL((k_sgemm_shortk_phased<true>), sgrid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_sgemm_shortk_phased<false>), sgrid, dim3(256), 0, st, a);
        LLMC_LAUNCH_CHECK();
        return LLMC_OK;
    }
    if (sgemm_wide_eligible(a, TA, TB)) return sgemm_wide_launch(a, st);
    LLMC_REQUIRE((const void*)a.C != (const void*)a.B || (a.M <= GB && a.batch == 1),
                 "sgemm: in-place C = op(A) B needs a single row tile (M <= 128)");
    dim3 grid((a.N + GB - 1) / GB, (a.M + GB - 1) / GB, a.batch);
    if (a.phase_len == 0 && a.epilogue == SG_SUB && !a.a_upper && !TB) {
        // plain C -= AB: one phase covering the whole K loop, i.e. the C tile is fetched while the first operand
        // tiles are, and the epilogue is stores only (same single rounding C - acc)
        SgemmArgs b = a;
        b.phase_len = 1 << 30;
        return sgemm_launch(b, TA, TB, st);
    }
    // interior-only instantiation when every tile and K range is whole (the shapes of the 128-aligned layers)
    auto whole = [&](int M, int N, int Kd) { return M % GB == 0 && N % GB == 0 && Kd % GK == 0; };
    const bool edge = !(whole(a.M, a.N, a.Kd) && whole(a.M_last, a.N_last, a.Kd_last));
#define LLMC_SG(TA_, TB_, PH_)                                                                        \
    do {                                                                                              \
        if (edge) hipLaunchKernelGGL((k_sgemm<TA_, TB_, PH_, true>), grid, dim3(256), 0, st, a);      \
        else hipLaunchKernelGGL((k_sgemm<TA_, TB_, PH_, false>), grid, dim3(256), 0, st, a);          \
    } while (0)
    if (a.phase_len > 0) {
        LLMC_REQUIRE(a.phase_len % GK == 0 && a.epilogue == SG_SUB && !a.a_upper, "sgemm: bad phased configuration");
        if (TA && !TB) LLMC_SG(true, false, true);
        else if (!TA && !TB) LLMC_SG(false, false, true);
        else { set_last_error_msg("sgemm: phased mode supports op(B) = N only"); return LLMC_ENOTSUP; }
    } else if (TA && !TB) LLMC_SG(true, false, false);
    else if (!TA && !TB) LLMC_SG(false, false, false);
    else if (TA && TB) LLMC_SG(true, true, false);
    else LLMC_SG(false, true, false);
#undef LLMC_SG
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

}  // namespace llmc

// C ABI test hook (not part of the product surface: exercised by tests/test_sgemm_gpu.py only)
extern "C" int llmc_test_sgemm(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc,
                               int M, int N, int Kd, int TA, int TB, int epilogue, int a_upper, int a_lower,
                               int b_upper, int c_upper_only, llmc_stream_t stream) {
    llmc::SgemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = a.M_last = M; a.N = a.N_last = N; a.Kd = a.Kd_last = Kd;
    a.epilogue = epilogue;
    a.a_upper = a_upper; a.a_lower = a_lower; a.b_upper = b_upper; a.c_upper_only = c_upper_only;
    a.batch = 1;
    return llmc::sgemm_launch(a, TA != 0, TB != 0, (hipStream_t)stream);
}
// the phased form (K4's far update): C -= op(A)[:, p] B[p, :] for p = phases of `phase_len` k, one launch
extern "C" int llmc_test_sgemm_phased(const float* A, const float* B, float* C, int64_t lda, int64_t ldb, int64_t ldc,
                                      int M, int N, int Kd, int TA, int phase_len, llmc_stream_t stream) {
    llmc::SgemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = a.M_last = M; a.N = a.N_last = N; a.Kd = a.Kd_last = Kd;
    a.epilogue = llmc::SG_SUB;
    a.batch = 1;
    a.phase_len = phase_len;
    return llmc::sgemm_launch(a, TA != 0, false, (hipStream_t)stream);
}
