// gptq_loop.hip — K4: GPTQ.weight_transform (gptq.py:199-244), the blocked column loop.
//
// Rows of W are independent given U (= Hinv, the upper factor), so the serial part is 128 dependent
// column steps per 128-column block. In-block kernel: a wave owns 4 rows, 16 lanes per row, lane p owns
// columns p, p+16, ..., p+112 of the block (interleaved so every lane has the same amount of trailing
// work at every step). Step i: the owner's current w_i is broadcast inside the 16-lane group
// (ds_swizzle), every lane evaluates the quantizer chain and err = (w - q) / d redundantly, then updates
// its own later columns  w_j <- w_j - round(err * U[i][j])  (two roundings, like the reference's
// `W1[:, i:] -= err1.unsqueeze(1).matmul(Hinv1[i, i:].unsqueeze(0))`). The U block lives in LDS in the
// lanes' ownership order with its diagonal and lower part zeroed, so no predicate is needed and each
// lane's registers end up holding exactly `tmp` (the weight of each column at the time it was visited).
// Trailing update W[:, i2:] -= Err1 @ U[i1:i2, i2:] runs on the fp32 MFMA pipe (sgemm.hip) with the
// k-ordered fma chain that reproduces the reference's CPU sgemm bit for bit.
#include <stdlib.h>
#include "common.h"
#include "quant_math.h"
#include "sgemm.h"
#include "side_stream.h"
#include "pipe_streams.h"

namespace llmc {

static constexpr int BS = 128;  // GPTQ blocksize

template <int PO> __device__ __forceinline__ float group_bcast(float v) {
    // lane' = (lane & 0x10) | PO inside each 32-lane half: broadcast of lane PO of every 16-lane group
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x10 | (PO << 5)));
}

struct GptqBlockArgs {
    const float* W;       // [R, K] running weights (panel read at cols i1..i1+count)
    const float* U;       // [K, K] upper factor
    float* Wout;          // [R, K] tmp
    float* losses;        // [R, K] or null
    float* Err;           // err of this block, c < 128: Err[row * err_ld + c], or k-major (err_kmajor) Err[c * err_ld + row]
    int err_ld;
    int err_kmajor;
    float* scales;        // [R, ng]
    float* zeros;         // [R, ng] or null (sym static)
    const int32_t* col_group;  // [K] group of processed column (static mode)
    int64_t R;
    int K;
    int i1;
    int count;            // columns in this block (<= 128)
    int ng;               // groups per row in scales/zeros
    int gsz;              // dynamic mode: group size (<= 128, divides 128); static mode: unused
    int static_mode;      // 0: qparams from current W at group starts; 1: given, gathered by col_group
    int sym;
    float qmin, qmax;
};

template <int I> struct StepIdx {
    static constexpr int PO = I & 15;
    static constexpr int EO = I >> 4;
};

// one column step, I compile-time
template <int I>
__device__ __forceinline__ void gptq_step(float (&w)[8], const float (&w0)[8], float (&er)[8], float (&ls)[8],
                                          float (&sc)[8], float (&zr)[8], const float* __restrict__ us,
                                          float d, int p, float& s_cur, float& z_cur, const GptqBlockArgs& a) {
    constexpr int PO = StepIdx<I>::PO, EO = StepIdx<I>::EO;
    // ---- group start (dynamic mode). The reference takes min/max from W[:, i:i+g] (gptq.py:216), which
    // inside a block still holds the values the block STARTED with (only the clone W1 receives the
    // in-block updates), hence w0 and not w for groups that start mid-block (group_size < 128).
    if (!a.static_mode && (I % 16 == 0)) {
        if ((I % a.gsz) == 0) {
            float mn = INFINITY, mx = -INFINITY;
            const int e1 = (I + a.gsz) >> 4;  // gsz is a multiple of 16
#pragma unroll
            for (int e = EO; e < 8; ++e)
                if (e < e1 && p + 16 * e < a.count) {
                    mn = fminf(mn, w0[e]);
                    mx = fmaxf(mx, w0[e]);
                }
            mn = wave_min(mn, 16);
            mx = wave_max(mx, 16);
            QParams q = qparams_from_minmax(mn, mx, LLMC_F32, a.sym, 1, a.qmin, a.qmax);
            s_cur = q.s;
            z_cur = q.z;
        }
    }
    float wi = group_bcast<PO>(w[EO]);
    float s = s_cur, z = z_cur;
    if (a.static_mode) {
        s = group_bcast<PO>(sc[EO]);
        z = group_bcast<PO>(zr[EO]);
    }
    const float qc = quant_code(wi, s, z, LLMC_F32, LLMC_F32, a.qmin, a.qmax);
    const float q = dequant_code(qc, s, z, LLMC_F32);
    const float diff = wi - q;
    const float err = diff / d;
    if (p == PO) {
        er[EO] = err;
        ls[EO] = (diff * diff) / (2.0f * (d * d));
    }
#pragma unroll
    for (int e = EO; e < 8; ++e) {
        const float u = us[I * BS + e];
        const float t = err * u;
        w[e] = w[e] - t;
    }
}

template <int I0>
__device__ __forceinline__ void gptq_steps16(float (&w)[8], const float (&w0)[8], float (&er)[8], float (&ls)[8],
                                             float (&sc)[8], float (&zr)[8], const float* __restrict__ us,
                                             const float* __restrict__ dg, int p, float& s_cur, float& z_cur,
                                             const GptqBlockArgs& a) {
#define LLMC_STEP(J)                                                                         \
    if (I0 + J < a.count) gptq_step<I0 + J>(w, w0, er, ls, sc, zr, us, dg[I0 + J], p, s_cur, z_cur, a);
    LLMC_STEP(0) LLMC_STEP(1) LLMC_STEP(2) LLMC_STEP(3) LLMC_STEP(4) LLMC_STEP(5) LLMC_STEP(6) LLMC_STEP(7)
    LLMC_STEP(8) LLMC_STEP(9) LLMC_STEP(10) LLMC_STEP(11) LLMC_STEP(12) LLMC_STEP(13) LLMC_STEP(14) LLMC_STEP(15)
#undef LLMC_STEP
}


// ---------------------------------------------------------------------------------------------------------
// Fast in-block path (count == 128). The serial chain of a column step is what bounds this kernel (one wave
// per SIMD at R = 4096), so the chain is cut to ~22 dependent VALU ops:
//   * the two IEEE divisions (w / scale and diff / d) divide by values that are fixed for many steps, so
//     the reciprocal refinement  y = rcp(d) * (2 - d * rcp(d))  is hoisted (per column for d, per group for
//     the scale) and each quotient is the remaining 5 ops of the very sequence hipcc emits for `n / d`
//     (mul, fma, fma, fma, div_fmas == fma).  That sequence first passes n and d through v_div_scale_f32,
//     which is the identity when both are "plain" (2^-40 <= |x| < 2^40, see the ISA's scaling rules), and
//     ends in v_div_fixup_f32, which only acts on zero / inf / nan / denormal operands.  Every numerator
//     of the block is still in registers after the loop (w[] holds each column's value at the time it was
//     visited, df[] each diff), so ONE check after the 128 steps proves all operands were plain (+0 counts:
//     the 5-op chain returns +0 for it, like the division); a wave that saw anything else (-0, tiny, huge,
//     inf, nan) discards its work and redoes the block with the generic path below, so results are
//     bit-identical by construction, not by argument.
//   * the broadcast of the current column is a DPP row_newbcast (VALU latency) instead of an LDS swizzle;
//   * the U row, d and 1/d of a step do not depend on the chain and are read from LDS ahead of it; there is
//     no control flow inside the 128 steps, the losses are evaluated after the loop.
template <int PO> __device__ __forceinline__ float row_bcast(float v) {
    // row_newbcast:PO (gfx90a+): every lane of a 16-lane row reads lane PO of its row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + PO, 0xf, 0xf, false));
}
// plain numerator: 2^-40 <= |x| < 2^40, or +0
__device__ __forceinline__ bool plain_num(float x) {
    const uint32_t b = __float_as_uint(x);
    return ((b & 0x7fffffffu) - 0x2B800000u) < 0x28000000u || b == 0u;
}
// n / d for plain d > 0 with y = rcp_refined(d) and n a plain numerator: the tail of hipcc's division sequence
__device__ __forceinline__ float div_plain(float n, float d, float y) {
    const float q0 = n * y;
    const float e1 = fmaf(-d, q0, n);
    const float q1 = fmaf(e1, y, q0);
    const float e2 = fmaf(-d, q1, n);
    return fmaf(e2, y, q1);
}

// One step. (u, dd) were loaded during the previous step; this step loads (un, ddn) for the next one first.
template <int I, bool STATIC>
__device__ __forceinline__ void fast_step(float (&w)[8], float (&er)[8], float (&df)[8], const float (&sc)[8],
                                          const float (&zr)[8], const float (&ys)[8],
                                          const float* __restrict__ us, const float2* __restrict__ dtab, int p,
                                          float s_cur, float z_cur, float y_cur, float qmin, float qmax,
                                          const float (&u)[8], const float2& dd, float (&un)[8],
                                          float2& ddn) {
    constexpr int PO = StepIdx<I>::PO, EO = StepIdx<I>::EO;
    if (I + 1 < BS) {
        constexpr int EN = StepIdx<I + 1>::EO;
#pragma unroll
        for (int e = EN; e < 8; ++e) un[e] = us[(I + 1) * BS + e];
        ddn = dtab[I + 1];
    }
    float s = s_cur, z = z_cur, y = y_cur;
    if (STATIC) {
        s = row_bcast<PO>(sc[EO]);
        z = row_bcast<PO>(zr[EO]);
        y = row_bcast<PO>(ys[EO]);
    }
    const float wi = row_bcast<PO>(w[EO]);
    float t = div_plain(wi, s, y);                  // quant_code(): x / s
    t = rintf(t);
    t = t + z;
    const float qc = fminf(fmaxf(t, qmin), qmax);
    const float q = (qc - z) * s;                   // dequant_code()
    const float diff = wi - q;
    const float err = div_plain(diff, dd.x, dd.y);
    const bool own = p == PO;
    er[EO] = own ? err : er[EO];
    df[EO] = own ? diff : df[EO];
    // pin the two selects here: left alone, the optimiser turns the 16-deep select chains into a private array
    // indexed by p after the loop, which keeps all 256 err / diff values alive (spills)
    asm volatile("" : "+v"(er[EO]), "+v"(df[EO]));
    // w[e] -= fl(err * u[e]) for e >= EO, two columns per packed instruction where a pair is whole
    typedef float v2f __attribute__((ext_vector_type(2)));
    if (EO & 1) {
        const float tt = err * u[EO];
        w[EO] = w[EO] - tt;
    }
#pragma unroll
    for (int e = (EO + 1) & ~1; e < 8; e += 2) {
        const v2f uu = {u[e], u[e + 1]};
        v2f ww = {w[e], w[e + 1]};
        const v2f tt = uu * err;
        ww = ww - tt;
        w[e] = ww.x;
        w[e + 1] = ww.y;
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting later steps' loads (register blow-up)
}

template <int I0, bool STATIC>
__device__ __forceinline__ void fast_steps16(float (&w)[8], float (&er)[8], float (&df)[8], const float (&sc)[8],
                                             const float (&zr)[8], const float (&ys)[8],
                                             const float* __restrict__ us, const float2* __restrict__ dtab,
                                             int p, float s_cur, float z_cur, float y_cur, float qmin,
                                             float qmax, float (&ua)[8], float2& da, float (&ub)[8],
                                             float2& db) {
#define LLMC_FSTEP2(J)                                                                                        \
    fast_step<I0 + J, STATIC>(w, er, df, sc, zr, ys, us, dtab, p, s_cur, z_cur, y_cur, qmin, qmax, ua, da, \
                              ub, db);                                                                        \
    fast_step<I0 + J + 1, STATIC>(w, er, df, sc, zr, ys, us, dtab, p, s_cur, z_cur, y_cur, qmin, qmax, ub, \
                                  db, ua, da);
    LLMC_FSTEP2(0) LLMC_FSTEP2(2) LLMC_FSTEP2(4) LLMC_FSTEP2(6) LLMC_FSTEP2(8) LLMC_FSTEP2(10) LLMC_FSTEP2(12)
    LLMC_FSTEP2(14)
#undef LLMC_FSTEP2
}

// Whole block for one wave (4 rows); returns false (and stores nothing) if any lane met a non-plain operand.
template <bool STATIC, int GSZ>
__device__ __forceinline__ bool block_fast(const GptqBlockArgs& a, const float* __restrict__ Us,
                                           const float2* __restrict__ dtab, int p, int64_t row, bool active) {
    const int64_t rr = active ? row : a.R - 1;
    float w[8], w0[8], er[8], df[8], sc[8], zr[8], ys[8];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = p + 16 * e;
        w[e] = a.W[rr * a.K + a.i1 + c];
        w0[e] = w[e];
        er[e] = 0.0f;
        df[e] = 0.0f;
        sc[e] = 1.0f;
        zr[e] = 0.0f;
        ys[e] = 1.0f;
        if (STATIC) {
            const int g = a.col_group ? a.col_group[a.i1 + c] : 0;
            sc[e] = a.scales[rr * a.ng + g];
            zr[e] = a.zeros ? a.zeros[rr * a.ng + g] : 0.0f;
            ys[e] = rcp_refined(sc[e]);
            bad |= !plain_pos(sc[e]);
        }
    }
    float s_cur = 1.0f, z_cur = 0.0f, y_cur = 1.0f;
    float s_grp[8], z_grp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s_grp[e] = 0.0f;
        z_grp[e] = 0.0f;
    }
    const float* us = Us + p * 8;
    float ua[8], ub[8];
    float2 da = dtab[0], db = da;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ua[e] = us[e];
        ub[e] = 0.0f;
    }
#define LLMC_FCHUNK(E)                                                                                 \
    if (!STATIC && ((16 * E) % GSZ) == 0) {                                                            \
        float mn = INFINITY, mx = -INFINITY;                                                           \
        constexpr int e1 = (16 * E + GSZ) >> 4;                                                        \
        _Pragma("unroll") for (int e = E; e < 8; ++e) if (e < e1) {                                    \
            mn = fminf(mn, w0[e]);                                                                     \
            mx = fmaxf(mx, w0[e]);                                                                     \
        }                                                                                              \
        mn = wave_min(mn, 16);                                                                         \
        mx = wave_max(mx, 16);                                                                         \
        const QParams qp = qparams_from_minmax(mn, mx, LLMC_F32, a.sym, 1, a.qmin, a.qmax);            \
        s_cur = qp.s;                                                                                  \
        z_cur = qp.z;                                                                                  \
        y_cur = rcp_refined(s_cur);                                                                    \
        bad |= !plain_pos(s_cur);                                                                      \
    }                                                                                                  \
    fast_steps16<16 * E, STATIC>(w, er, df, sc, zr, ys, us, dtab, p, s_cur, z_cur, y_cur, a.qmin, a.qmax, \
                                 ua, da, ub, db);                                                      \
    s_grp[E] = s_cur;                                                                                  \
    z_grp[E] = z_cur;
    LLMC_FCHUNK(0) LLMC_FCHUNK(1) LLMC_FCHUNK(2) LLMC_FCHUNK(3) LLMC_FCHUNK(4) LLMC_FCHUNK(5) LLMC_FCHUNK(6)
    LLMC_FCHUNK(7)
#undef LLMC_FCHUNK
#pragma unroll
    for (int e = 0; e < 8; ++e) bad |= !plain_num(w[e]) | !plain_num(df[e]);
    if (__any(bad)) return false;
    if (!active) return true;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = p + 16 * e;
        a.Wout[row * a.K + a.i1 + c] = w[e];
        if (a.losses) {
            const float d = dtab[c].x;
            a.losses[row * a.K + a.i1 + c] = (df[e] * df[e]) / (2.0f * (d * d));
        }
        a.Err[a.err_kmajor ? (int64_t)c * a.err_ld + row : (int64_t)row * a.err_ld + c] = er[e];
    }
    if (!STATIC && p == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = 16 * e;
            if ((i % GSZ) == 0) {
                const int g = (a.i1 + i) / GSZ;
                a.scales[row * a.ng + g] = s_grp[e];
                if (a.zeros) a.zeros[row * a.ng + g] = z_grp[e];
            }
        }
    }
    return true;
}

static constexpr int GBT = 512;  // threads per workgroup: 8 waves x 4 rows (1024 for tall weights, see launch)

// VARIANT: 0 generic path only; 1 fast path for given qparams (static groups / per-channel); 16/32/64/128 fast
// path for qparams taken at group starts with that group size. The fast variants fall back to the generic code
// per wave.
template <int VARIANT, int NT>
__global__ __launch_bounds__(NT) void k_gptq_block(GptqBlockArgs a) {
    // Us[i][p*8 + e] = U[i1+i][i1 + p + 16e] for p+16e > i, else 0 ; dg[i] = U[i1+i][i1+i]
    __shared__ __attribute__((aligned(16))) float Us[BS * BS];
    __shared__ float dg[BS];
    __shared__ float2 dtab[BS];   // fast path: {d, refined 1/d}
    __shared__ int d_not_plain;   // some d of the block is outside the plain range: generic path for everyone
    if (threadIdx.x == 0) d_not_plain = 0;
    __syncthreads();
    const int tid = threadIdx.x;
    {
        // all 8 float4 loads of a thread are issued before the first use (a per-element loop serialises 32
        // L2 round trips, which used to be most of this kernel's time)
        constexpr int NV = BS * BS / 4 / NT;
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + NT * j;
            const int i = idx >> 5, c4 = (idx & 31) * 4;   // count and i1 are multiples of 4
            v[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (i < a.count && c4 < a.count)
                v[j] = *reinterpret_cast<const float4*>(a.U + (int64_t)(a.i1 + i) * a.K + a.i1 + c4);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + NT * j;
            const int i = idx >> 5, c4 = (idx & 31) * 4;
            const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = c4 + t;   // column inside the block
                if (c == i && i < a.count) {
                    dg[i] = vv[t];
                    dtab[i] = make_float2(vv[t], rcp_refined(vv[t]));
                    if (!plain_pos(vv[t])) d_not_plain = 1;
                }
                Us[i * BS + (c & 15) * 8 + (c >> 4)] = c > i ? vv[t] : 0.0f;
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int p = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 4 + (lane >> 4);
    const bool active = row < a.R;
    const int64_t rr = active ? row : a.R - 1;

    if (VARIANT != 0 && !d_not_plain) {
        const bool done = block_fast<VARIANT == 1, VARIANT == 1 ? BS : VARIANT>(a, Us, dtab, p, row, active);
        if (done) return;
    }

    float w[8], w0[8], er[8], ls[8], sc[8], zr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = p + 16 * e;
        w[e] = (c < a.count) ? a.W[rr * a.K + a.i1 + c] : 0.0f;
        w0[e] = w[e];
        er[e] = 0.0f;
        ls[e] = 0.0f;
        sc[e] = 1.0f;
        zr[e] = 0.0f;
        if (a.static_mode && c < a.count) {
            const int g = a.col_group ? a.col_group[a.i1 + c] : 0;
            sc[e] = a.scales[rr * a.ng + g];
            zr[e] = a.zeros ? a.zeros[rr * a.ng + g] : 0.0f;
        }
    }
    float s_cur = 1.0f, z_cur = 0.0f;
    float s_grp[8], z_grp[8];  // dynamic mode: qparams captured at each 16-column boundary (group starts)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s_grp[e] = 0.0f;
        z_grp[e] = 0.0f;
    }
    const float* us = Us + p * 8;

#define LLMC_CHUNK(E)                                                             \
    gptq_steps16<16 * E>(w, w0, er, ls, sc, zr, us, dg, p, s_cur, z_cur, a);  \
    s_grp[E] = s_cur;                                                             \
    z_grp[E] = z_cur;
    LLMC_CHUNK(0) LLMC_CHUNK(1) LLMC_CHUNK(2) LLMC_CHUNK(3) LLMC_CHUNK(4) LLMC_CHUNK(5) LLMC_CHUNK(6) LLMC_CHUNK(7)
#undef LLMC_CHUNK

    if (!active) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = p + 16 * e;
        if (c < a.count) {
            a.Wout[row * a.K + a.i1 + c] = w[e];
            if (a.losses) a.losses[row * a.K + a.i1 + c] = ls[e];
        }
        a.Err[a.err_kmajor ? (int64_t)c * a.err_ld + row : (int64_t)row * a.err_ld + c] = (c < a.count) ? er[e] : 0.0f;
    }
    if (!a.static_mode && p == 0) {
        // qparams of the groups that start in this block (gsz divides 128, multiple of 16)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = 16 * e;
            if (i < a.count && (i % a.gsz) == 0) {
                const int g = (a.i1 + i) / a.gsz;
                a.scales[row * a.ng + g] = s_grp[e];
                if (a.zeros) a.zeros[row * a.ng + g] = z_grp[e];
            }
        }
    }
}

}  // namespace llmc

using namespace llmc;

static constexpr int GRP = 4;  // 128-column blocks per outer group (far updates are applied once per group; 8 measured the same: 12.4 vs 12.6 ms)

extern "C" size_t llmc_gptq_quantize_ws_bytes(int64_t R, int64_t K) {
    if (R <= 0 || K <= 0) return 0;
    return 3 * (size_t)((R + 3) & ~(int64_t)3) * BS * GRP * sizeof(float);   // err columns of three groups in flight (pipelined far updates)
}

extern "C" int llmc_gptq_quantize(float* W, const float* Hinv, int64_t R, int64_t K, int sym, float qmin,
                                  float qmax, int64_t group_size, int static_groups, const int32_t* col_group,
                                  float* scales, float* zeros, float* Wout, float* losses, int blocksize,
                                  void* ws, llmc_stream_t stream) {
    return llmc_gptq_quantize_cols(W, Hinv, R, K, K, sym, qmin, qmax, group_size, static_groups, col_group, scales,
                                   zeros, Wout, losses, blocksize, ws, stream);
}

// OWQ form (gptq.py:44-56,199-244 with n_nonout < columns): only the first n_quant columns are visited by the column
// loop; the trailing K - n_quant columns (the outlier columns OWQ keeps in floating point) still receive every
// block's error feedback `W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]`. Groups are clipped at n_quant like the reference's
// `min(i + group_size, columns - n_out)`.
extern "C" int llmc_gptq_quantize_cols(float* W, const float* Hinv, int64_t R, int64_t K, int64_t n_quant, int sym,
                                       float qmin, float qmax, int64_t group_size, int static_groups,
                                       const int32_t* col_group, float* scales, float* zeros, float* Wout,
                                       float* losses, int blocksize, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(W && Hinv && Wout && scales && ws && R > 0 && K > 0, "gptq_quantize: null/empty argument");
    LLMC_REQUIRE(n_quant > 0 && n_quant <= K, "gptq_quantize: n_quant must be in (0, K]");
    const int64_t NQ = n_quant;
    LLMC_REQUIRE(blocksize == BS, "gptq_quantize: blocksize must be 128");
    LLMC_REQUIRE(K % 4 == 0 && K < (1 << 30), "gptq_quantize: K must be a multiple of 4");
    LLMC_REQUIRE(sym || zeros, "gptq_quantize: zeros required for asymmetric");
    const bool per_channel = group_size <= 0;
    int static_mode = static_groups || per_channel;
    int gsz = (int)group_size;
    int ng = per_channel ? 1 : (int)ceil_div64(K, group_size);
    if (!static_mode) {
        if (!(gsz == 16 || gsz == 32 || gsz == 64 || gsz == 128)) {
            set_last_error_msg("gptq_quantize: dynamic group qparams need group_size in {16,32,64,128}");
            return LLMC_ENOTSUP;
        }
    } else if (!per_channel) {
        LLMC_REQUIRE(col_group != nullptr, "gptq_quantize: col_group required with static groups");
    }
    hipStream_t caller = (hipStream_t)stream;
    const int ELD = BS * GRP;
    // Round 5: the error columns are kept K-MAJOR, [GRP * 128][Rp] — the fp32 GEMMs stage a k-major A panel with 16-B LDS writes,
    // a row-major one ([R][512]) through four scalar transposing writes per float4 (square 4096: 130 vs 122 TFLOP/s). The in-block
    // kernel's stores become 16-B segments (four rows of one column); the volume is 2 MB per block. option k4_err_rowmajor: the old
    // layout (same bits).
    const bool ekm = !opt(OPT_K4_ERR_ROWMAJOR);
    const int64_t Rp = (R + 3) & ~(int64_t)3;
    const int64_t err_ld = ekm ? Rp : ELD;
    float* ErrBuf[3] = {(float*)ws, (float*)ws + (size_t)Rp * ELD, (float*)ws + 2 * (size_t)Rp * ELD};   // [R, GRP*128] or [GRP*128, Rp], x 3
    // Round 4 schedule. The caller's stream carries the CHAIN: per 128-column block the in-block kernel and the update of the
    // rest of its group's columns, per group the update of the NEXT group's columns. The `bulk` helper stream (CU-masked:
    // pipe_streams.h) carries the update of everything beyond the next group, the columns of the group after next FIRST
    // (event C1: all the next group's own update waits for), then the rest. Round 3 joined the whole side update before
    // every group's far update, so the chain stood still while ~430 us of fp32 far update drained. Per element the updates
    // still arrive in the reference's order (block 0, 1, 2, ...: bulk stream order, then the chain behind C1), from the
    // same kernels on the same tile grid: bit-identical to one stream (llmc_hip_set_helper_streams(0)), which tests compare.
    PipeStreams* ps = (!helper_streams_enabled()) ? nullptr : pipe_streams_for(caller);
    const bool merge_far = !opt(OPT_K4_SPLIT_FAR);
    hipStream_t st = ps ? pipe_chain_stream(ps, caller) : caller;
    hipStream_t bulk = ps ? ps->bulk : st;
    if (ps) {
        hipEvent_t e0 = nullptr;
        int rc = ps->record(caller, &e0);
        if (rc) return rc;
        if (st != caller && (rc = pipe_wait(st, e0))) return rc;
        if ((rc = pipe_wait(bulk, e0))) return rc;
    }
    hipEvent_t C1_prev = nullptr;                 // the previous group's far-far update has reached the next group's columns
    hipEvent_t C2_hist[3] = {nullptr, nullptr, nullptr};   // ... is complete (its err buffer may be rewritten)
    hipEvent_t bulk_tail = nullptr;               // behind the most recent launch on the bulk stream
    const int force_generic = opt(OPT_GPTQ_GENERIC) ? 1 : 0;
    // Every weight receives the blocks' updates in the reference's order (block 0, 1, 2, ...), each as
    // "W -= chain over the block's 128 k" (gptq.py:244). Columns inside the current outer group get them right
    // after each block (the next block needs them); columns beyond the group get the group's GRP updates in one
    // phased GEMM that keeps the C tile in registers — same arithmetic, one pass over the far columns per group.
    int gidx = 0;
    for (int64_t g0 = 0; g0 < NQ; g0 += (int64_t)BS * GRP, ++gidx) {
        const int64_t gend = g0 + (int64_t)BS * GRP < NQ ? g0 + (int64_t)BS * GRP : NQ;
        // columns updated right after every block: up to the end of the outer group; in the LAST group also the
        // never-visited columns beyond n_quant (their group-wide phased update could start on a ragged phase)
        const int64_t near_end = gend == NQ ? K : gend;
        float* Err = ErrBuf[gidx % 3];
        if (ps && near_end > gend) {
            // OWQ's last group: its per-block updates reach the never-visited columns [n_quant, K), which the earlier
            // groups' far-far updates on the bulk stream also write (nothing else orders the two when
            // last_group_start + 512 < K). Everything queued on the bulk stream so far has to land first.
            int rc = pipe_wait(st, bulk_tail);
            if (rc) return rc;
        }
        if (ps && C2_hist[gidx % 3]) {            // the far-far update that read this err buffer three groups ago
            int rc = pipe_wait(st, C2_hist[gidx % 3]);
            if (rc) return rc;
        }
        for (int64_t i1 = g0; i1 < gend; i1 += BS) {
            const int count = (int)(NQ - i1 < BS ? NQ - i1 : BS);
            GptqBlockArgs a;
            a.W = W; a.U = Hinv; a.Wout = Wout; a.losses = losses;
            a.Err = Err + (ekm ? (i1 - g0) * Rp : (i1 - g0)); a.err_ld = (int)err_ld; a.err_kmajor = ekm ? 1 : 0;
            a.scales = scales; a.zeros = zeros; a.col_group = per_channel ? nullptr : col_group;
            a.R = R; a.K = (int)K; a.i1 = (int)i1; a.count = count; a.ng = ng; a.gsz = static_mode ? BS : gsz;
            a.static_mode = static_mode; a.sym = sym; a.qmin = qmin; a.qmax = qmax;
            // 64 KB of LDS per workgroup = 2 workgroups per CU: tall weights use 1024-thread workgroups so that the
            // whole grid is resident at once (R = 28672: 448 workgroups on 512 slots instead of 896)
            const int nt = R >= 16384 ? 1024 : GBT;
            const int grid = (int)ceil_div64(R, nt / 16);
            // group sizes 16/32/64 with qparams taken mid-block stay on the generic path (their fast variants
            // spill: the qparams change inside the unrolled loop)
            const int variant = (count != BS || force_generic) ? 0 : static_mode ? 1 : gsz == BS ? BS : 0;
            switch (variant) {
#define LLMC_GB(V)                                                                                   \
    case V:                                                                                          \
        if (nt == 1024) hipLaunchKernelGGL((k_gptq_block<V, 1024>), dim3(grid), dim3(1024), 0, st, a); \
        else hipLaunchKernelGGL((k_gptq_block<V, GBT>), dim3(grid), dim3(GBT), 0, st, a);             \
        break;
                LLMC_GB(0) LLMC_GB(1) LLMC_GB(128)
#undef LLMC_GB
            }
            LLMC_LAUNCH_CHECK();
            const int64_t i2 = i1 + count;
            if (i2 < near_end) {   // near columns of the group
                // the GEMM wants 16-B aligned operands: a ragged n_quant (OWQ) starts up to 3 columns early, on
                // columns the loop has already visited — W is dead there (their values live in Wout)
                const int64_t c0 = i2 & ~(int64_t)3;
                SgemmArgs g{};
                g.A = Err + (ekm ? (i1 - g0) * Rp : (i1 - g0)); g.lda = err_ld;
                g.B = Hinv + i1 * K + c0; g.ldb = K;
                g.C = W + c0; g.ldc = K;
                g.M = g.M_last = (int)R; g.N = g.N_last = (int)(near_end - c0); g.Kd = g.Kd_last = count;
                g.epilogue = SG_SUB; g.batch = 1;
                int rc = sgemm_launch(g, ekm, false, st);
                if (rc) return rc;
            }
        }
        if (near_end < K) {    // far columns: GRP phases of 128
            const int64_t gend2 = gend + (int64_t)BS * GRP < K ? gend + (int64_t)BS * GRP : K;
            // the next group's columns were last written by the previous group's far-far update (its first part)
            if (ps) {
                int rc = pipe_wait(st, C1_prev);
                if (rc) return rc;
            }
            SgemmArgs g{};
            g.A = Err; g.lda = err_ld;
            g.B = Hinv + g0 * K + gend; g.ldb = K;
            g.C = W + gend; g.ldc = K;
            g.M = g.M_last = (int)R; g.N = g.N_last = (int)(gend2 - gend); g.Kd = g.Kd_last = (int)(gend - g0);
            g.epilogue = SG_SUB; g.batch = 1; g.phase_len = BS;
            // Without helper streams the three column ranges of the group's far update (the next group's columns, the group
            // after next, the rest) are one product on one stream: ONE launch over [gend, K). Column tiles are independent and
            // every kernel the GEMM may pick computes an element the same way (one accumulator per phase from +0 in ascending k,
            // then one rounding C - acc): same bits, two launches less per group and no half-empty 128-workgroup grids.
            const bool one_far = !ps && merge_far;
            if (one_far) g.N = g.N_last = (int)(K - gend);
            int rc = sgemm_launch(g, ekm, false, st);
            if (rc) return rc;
            C1_prev = nullptr;
            if (gend2 < K && !one_far) {
                if (ps) {
                    hipEvent_t e = nullptr;       // this group's err columns are complete on the chain
                    if ((rc = ps->record(st, &e))) return rc;
                    if ((rc = pipe_wait(bulk, e))) return rc;
                }
                // first the columns of the group after next, then the rest (same tiles as one launch: column tiles are
                // independent and both cuts are multiples of the tile width)
                const int64_t gend3 = gend2 + (int64_t)BS * GRP < K ? gend2 + (int64_t)BS * GRP : K;
                SgemmArgs h = g;
                h.B = Hinv + g0 * K + gend2;
                h.C = W + gend2;
                h.N = h.N_last = (int)(gend3 - gend2);
                if ((rc = sgemm_launch(h, ekm, false, bulk))) return rc;
                if (ps && (rc = ps->record(bulk, &C1_prev))) return rc;
                if (gend3 < K) {
                    SgemmArgs h2 = g;
                    h2.B = Hinv + g0 * K + gend3;
                    h2.C = W + gend3;
                    h2.N = h2.N_last = (int)(K - gend3);
                    if ((rc = sgemm_launch(h2, ekm, false, bulk))) return rc;
                }
                if (ps && (rc = ps->record(bulk, &C2_hist[gidx % 3]))) return rc;
                bulk_tail = C2_hist[gidx % 3];
            }
        }
    }
    if (ps) {
        hipEvent_t e = nullptr;
        int rc = ps->record(bulk, &e);
        if (rc) return rc;
        if ((rc = pipe_wait(st, e))) return rc;
        if (st != caller) {
            if ((rc = ps->record(st, &e))) return rc;
            if ((rc = pipe_wait(caller, e))) return rc;
        }
    }
    return LLMC_OK;
}
