// awq_clip.hip — K10: AutoClipper.auto_clip_layer (auto_clip.py:84-191), clip_version v1, weight-only.
//
// For every (output row, input group) the reference evaluates 10 shrink levels of the group's clipping range:
//   q_w = fakequant_dyn(clamp(w, -m, m)),  err = mean_tok((sum_k x*q_w - sum_k x*w)^2),  keep the argmin,
// with every op in the model dtype: each product x*w is rounded to 16 bit BEFORE the k-sum, so this is not an
// MFMA contraction (the matrix pipe does not round products) — it is evaluated on the VALU with the
// reference's roundings, which is cheap enough (2*11*R*K*n_tok rounded MACs, ~10 ms for 4096^2 x 512 tokens)
// and keeps the chosen clip levels identical to the reference's.
// Layout: one workgroup per (group, slab of rows). The group's activations [n_tok, 128] are staged once,
// transposed to [k][token] in LDS (lane <-> token: conflict-free ds_read_b64 of 4 tokens); each wave walks
// rows; the 11 candidate weight vectors of a row (original + 10 clipped/fake-quantized) live in a small LDS
// table read as broadcasts.
#include "common.h"
#include "quant_math.h"

namespace llmc {

static constexpr int CG = 128;        // max group size handled per LDS column block
static constexpr int CTOK = 512;      // tokens per LDS tile
static constexpr int CROWS = 32;      // rows per workgroup
static constexpr int CMAXS = 12;      // 1 + max shrink steps (n_shrink <= 11)

template <int DT> __device__ __forceinline__ float rfast(float v) {
    if constexpr (DT == LLMC_F16) {
        return f16_bits_to_f32(f32_to_f16_bits(v));     // through the fp32 value (no v_fma_mixlo_f16 fusion, common.h)
    } else if constexpr (DT == LLMC_BF16) {
        uint32_t x = __float_as_uint(v);
        x += 0x7fffu + ((x >> 16) & 1u);
        return __uint_as_float(x & 0xffff0000u);
    } else {
        return v;
    }
}

// round four fp32 values to DT and back (bf16: two v_cvt_pk_bf16_f32 + shifts/masks instead of 16 integer ops)
template <int DT> __device__ __forceinline__ void rfast4(float (&v)[4]) {
    if constexpr (DT == LLMC_BF16) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 a = {v[0], v[1]}, b = {v[2], v[3]};
        const uint32_t pa = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2));
        const uint32_t pb = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf2));
        v[0] = __uint_as_float(pa << 16);
        v[1] = __uint_as_float(pa & 0xffff0000u);
        v[2] = __uint_as_float(pb << 16);
        v[3] = __uint_as_float(pb & 0xffff0000u);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rfast<DT>(v[j]);
    }
}

struct ClipArgs {
    const void* W;   // [R, K]
    const void* X;   // [n_tok, K]
    int64_t R, K;
    int g, ng, n_tok;
    int n_grid, n_shrink, clip_sym, sym;
    float qmin, qmax;
    void* best_max;  // [R, ng]
    void* best_min;
    void* errs;      // optional [n_shrink, R, ng] dt: the mean error of every shrink level (multi-batch callers)
};

// ---- ATen's summation orders (the reference's CPU path, the one the goldens come from) ------------------------------
// The chosen clip level is an argmin over errors that are 16-bit numbers: one ulp of difference in one output flips it.
// Sums of 16-bit values in fp32 are not order-independent in their last bits (measured on the goldens: a plain
// sequential k-sum changes 1-2.4 % of the fp16 error entries), so both reductions follow ATen
// (aten/src/ATen/native/cpu/SumKernel.cpp; sum_stub has no AVX-512 variant, so the AVX2 kernel runs on every x86 host:
// V = 8 fp32 lanes; restated and pinned against torch itself in oracle/aten_sum.py):
//  (1) `(x * w).sum(-1)` (auto_clip.py:150,166): vectorized_inner_sum with the reduced-precision load policy: a 16-element
//      Vectorized<BFloat16 / Half> is loaded as lo(8) + hi(8) in fp32; row_sum over the row's vectors: four interleaved
//      streams (vector i -> stream i % 4, the vectors past the last full four -> stream 0), ((s0 + s1) + s2) + s3 per
//      lane; the 8 lane sums added sequentially, trailing elements (g % 16) first. One thread per output: independent
//      of the host's thread count.
//  (2) `.pow(2).mean(dim=1)` (auto_clip.py:170): 16-bit means are cast to fp32, summed and divided; the sum over tokens is
//      cascade_sum's outer reduction for the serial iterator (what small inputs and the goldens take; large inputs are
//      split across threads by TensorIterator, there the reference's own order depends on the thread count):
//      multi_row_sum — 16-element chunks summed sequentially, chunk sums cascaded in levels of 16 — either directly
//      (MODE_A) or on four interleaved streams i % 4 that are added at the end (row_sum, MODE_B); which one a column
//      (= group index j of ng) gets is vectorized_outer_sum's / scalar_outer_sum's blocking by 32 / 8 / 4 columns.
// oracle/awq_ref.py pins the resulting levels 100 % to the reference on tests/golden/clip.npz and clip_mb.npz.
__device__ __forceinline__ bool aten_outer_mode_b(int ng, int j) {
    if (ng >= 8) return j >= (ng / 32) * 32;       // 4 vectors of 8 columns at a time: multi_row_sum; the rest: row_sum
    return j >= (ng / 4) * 4;                      // scalar_outer_sum: 4 columns at a time, the rest row_sum
}

// fp32 sum of v[0 .. n) (LDS, n <= 512) in ATen's cascade order, by one wave; every lane returns the result.
// stride 1 / 4: the elements i, i + stride, ... of one interleaved stream (MODE_B calls it with stride 4 per stream).
__device__ __forceinline__ float aten_cascade_sum(const float* v, int first, int stride, int count, int lane) {
    // chunk sums: lane c sums elements [16 c, 16 c + 16) of the stream sequentially from 0
    const int nch = count >> 4;
    float cs = 0.f;
    if (lane < nch) {
#pragma unroll
        for (int m = 0; m < 16; ++m) cs += v[first + (16 * lane + m) * stride];
    }
    // cascade over the chunk sums (level_step 16: acc1 takes 16 chunk sums, then moves on to acc2), remainder into acc0
    float acc1 = 0.f, acc2 = 0.f;
    for (int c = 0; c < nch; ++c) {
        acc1 += __shfl(cs, c, 64);
        if (((c + 1) & 15) == 0) {
            acc2 += acc1;
            acc1 = 0.f;
        }
    }
    float acc0 = 0.f;
    for (int i = nch * 16; i < count; ++i) acc0 += v[first + i * stride];
    acc0 += acc1;
    acc0 += acc2;
    return acc0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_clip_search(ClipArgs a) {
    constexpr int DT = dt_of<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xt = (T*)smem;                                                  // [g][CTOK]
    T* tab = (T*)(smem + (size_t)CG * CTOK * sizeof(T));               // [4 waves][CMAXS][CG] candidate weights (dt values)
    float* esum_all = (float*)(tab + 4 * CMAXS * CG);                  // [CROWS][CMAXS]
    float* sqbuf = esum_all + CROWS * CMAXS;                           // [4 waves][CTOK] squared errors of one (row, level)
    const int gi = blockIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * CROWS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = a.g;
    T* mytab = tab + wv * CMAXS * CG;
    float* mysq = sqbuf + wv * CTOK;
    const int ns = a.n_shrink;
    const bool v0 = lane < g, v1 = lane + 64 < g;
    const bool exact_tok = a.n_tok <= CTOK;           // one tile: the token sum follows ATen's serial cascade
    const bool mode_b = aten_outer_mode_b(a.ng, gi);
    const int nv = g >> 4;                            // 16-element vectors of the k-sum (<= 8)
    const int nv4 = nv & ~3;                          // vectors dealt to the four streams; the rest go to stream 0

    for (int e = tid; e < CROWS * CMAXS; e += 256) esum_all[e] = 0.f;

    for (int t0 = 0; t0 < a.n_tok; t0 += CTOK) {
        const int nt = a.n_tok - t0 < CTOK ? a.n_tok - t0 : CTOK;
        __syncthreads();   // previous tile fully consumed by every wave (and esum_all zeroed)
        for (int e = tid; e < CTOK * g; e += 256) {   // stage X[t0:t0+nt, group] transposed to [k][token]
            const int t = e / g, k = e - t * g;
            T v = from_f32<T>(0.f);
            if (t < nt) v = ((const T*)a.X)[(int64_t)(t0 + t) * a.K + (int64_t)gi * g + k];
            xt[k * CTOK + t] = v;
        }
        __syncthreads();
        for (int rr = wv; rr < CROWS; rr += 4) {   // no barrier inside: waves walk their own rows
            const int64_t row = r0 + rr;
            if (row >= a.R) continue;
            // ---- the 1 + ns candidate weight vectors of (row, group) -> this wave's LDS table
            const T* wp = (const T*)a.W + row * a.K + (int64_t)gi * g;
            const float w0 = v0 ? to_f32<T>(wp[lane]) : 0.f, w1 = v1 ? to_f32<T>(wp[lane + 64]) : 0.f;
            const float amx = fmaxf(v0 ? (a.clip_sym ? fabsf(w0) : w0) : -INFINITY,
                                    v1 ? (a.clip_sym ? fabsf(w1) : w1) : -INFINITY);
            const float amn = fminf(v0 ? w0 : INFINITY, v1 ? w1 : INFINITY);
            const float org_max = wave_max(amx, 64), org_min = wave_min(amn, 64);
            mytab[lane] = from_f32<T>(w0);
            mytab[lane + 64] = from_f32<T>(w1);
            for (int s = 0; s < ns; ++s) {
                const float f = (float)(1.0 - (double)s / (double)a.n_grid);   // python scalar: fp32 opmath in ATen's mul
                const float mx = rndc<DT>(org_max * f);
                const float mn = a.clip_sym ? -mx : rndc<DT>(org_min * f);
                const float c0 = fminf(fmaxf(w0, mn), mx), c1 = fminf(fmaxf(w1, mn), mx);
                const float gmx = wave_max(fmaxf(v0 ? c0 : -INFINITY, v1 ? c1 : -INFINITY), 64);
                const float gmn = wave_min(fminf(v0 ? c0 : INFINITY, v1 ? c1 : INFINITY), 64);
                const QParams q = qparams_from_minmax(gmn, gmx, DT, a.sym, 1, a.qmin, a.qmax);
                const float q0 = dequant_code(quant_code(c0, q.s, q.z, DT, DT, a.qmin, a.qmax), q.s, q.z, DT);
                const float q1 = dequant_code(quant_code(c1, q.s, q.z, DT, DT, a.qmin, a.qmax), q.s, q.z, DT);
                mytab[(s + 1) * CG + lane] = from_f32<T>(v0 ? q0 : 0.f);
                mytab[(s + 1) * CG + lane + 64] = from_f32<T>(v1 ? q1 : 0.f);
            }
            // ---- outputs of every candidate for this tile's tokens: 4 tokens per lane, 256 per pass; candidate by
            // candidate (runtime loop: its 4 x 16 lane-partials of the ATen k-sum stay in registers)
            float o0[CTOK / 256][4];                 // the original weights' outputs (candidate 0)
            for (int s = 0; s <= ns; ++s) {
                const T* tw = mytab + s * CG;
                float es_lane = 0.f;                 // more than one tile: plain per-lane partial sums
#pragma unroll
                for (int pass = 0; pass < CTOK / 256; ++pass) {
                    const int tb = pass * 256;
                    if (tb >= nt) continue;
                    const int tl = tb + lane * 4;
                    float part[4][4][8];             // [stream][token][lane of the fp32 vector]
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int l = 0; l < 8; ++l) part[st][j][l] = 0.f;
#pragma unroll
                    for (int i = 0; i < CG / 16; ++i) {
                        if (i < nv) {                                    // uniform
#pragma unroll
                            for (int l = 0; l < 8; ++l) {
                                const int k1 = 16 * i + l, k2 = k1 + 8;
                                const float wa = to_f32<T>(tw[k1]), wb = to_f32<T>(tw[k2]);
                                float xa[4], xb[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    xa[j] = to_f32<T>(xt[k1 * CTOK + tl + j]) * wa;
                                    xb[j] = to_f32<T>(xt[k2 * CTOK + tl + j]) * wb;
                                }
                                rfast4<DT>(xa);
                                rfast4<DT>(xb);
                                if (i < nv4) {                           // lo + hi, then onto the stream's lane sum
#pragma unroll
                                    for (int j = 0; j < 4; ++j) part[i & 3][j][l] += xa[j] + xb[j];
                                } else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) part[0][j][l] += xa[j] + xb[j];
                                }
                            }
                        }
                    }
                    float fin[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int k = nv * 16; k < g; ++k) {           // trailing elements first
                        const float wk = to_f32<T>(tw[k]);
                        float xk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) xk[j] = to_f32<T>(xt[k * CTOK + tl + j]) * wk;
                        rfast4<DT>(xk);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fin[j] += xk[j];
                    }
#pragma unroll
                    for (int l = 0; l < 8; ++l)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            fin[j] += ((part[0][j][l] + part[1][j][l]) + part[2][j][l]) + part[3][j][l];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float o = rfast<DT>(fin[j]);
                        if (s == 0) {
                            o0[pass][j] = o;
                        } else {
                            const float d = rfast<DT>(o - o0[pass][j]);
                            const float sq = rfast<DT>(d * d);
                            if (tl + j < nt) {
                                mysq[tl + j] = sq;
                                es_lane += sq;
                            }
                        }
                    }
                }
                if (s == 0) continue;
                float tot;
                if (exact_tok) {
                    // the wave's own LDS writes are visible to its own later reads (in-order LDS queue per wave)
                    if (!mode_b) {
                        tot = aten_cascade_sum(mysq, 0, 1, nt, lane);
                    } else {                                   // row_sum: 4 interleaved streams + the n % 4 tail
                        const int rows4 = nt >> 2;
                        float part[4];
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) part[k4] = aten_cascade_sum(mysq, k4, 4, rows4, lane);
                        for (int i = rows4 * 4; i < nt; ++i) part[0] += mysq[i];
                        tot = ((part[0] + part[1]) + part[2]) + part[3];
                    }
                    if (lane == 0) esum_all[rr * CMAXS + s] = tot;
                } else {
                    tot = wave_sum(es_lane, 64);
                    if (lane == 0) esum_all[rr * CMAXS + s] += tot;
                }
            }
        }
    }
    __syncthreads();
    // ---- argmin over shrink levels (strict <, first minimum wins; min_errs starts at 1e9 in dt)
    for (int rr = tid; rr < CROWS; rr += 256) {
        const int64_t row = r0 + rr;
        if (row >= a.R) continue;
        const T* wp = (const T*)a.W + row * a.K + (int64_t)gi * g;
        float org_max = -INFINITY, org_min = INFINITY;
        for (int k = 0; k < g; ++k) {
            const float w = to_f32<T>(wp[k]);
            org_max = fmaxf(org_max, a.clip_sym ? fabsf(w) : w);
            org_min = fminf(org_min, w);
        }
        float best_mx = org_max, best_mn = org_min;
        float min_err = rndc<DT>(1e9f);
        for (int s = 0; s < ns; ++s) {
            const float e = rndc<DT>(esum_all[rr * CMAXS + s + 1] / (float)a.n_tok);
            if (a.errs) ((T*)a.errs)[((int64_t)s * a.R + row) * a.ng + gi] = from_f32<T>(e);
            const float f = (float)(1.0 - (double)s / (double)a.n_grid);
            const float mx = rndc<DT>(org_max * f);
            const float mn = a.clip_sym ? -mx : rndc<DT>(org_min * f);
            if (e < min_err) {
                min_err = e;
                best_mx = mx;
                best_mn = mn;
            }
        }
        if (a.best_max) ((T*)a.best_max)[row * a.ng + gi] = from_f32<T>(best_mx);
        if (a.best_min) ((T*)a.best_min)[row * a.ng + gi] = from_f32<T>(best_mn);
    }
}


// ---- K10w: error table from GIVEN candidates — every group width, every quantizer ----------------------------------
// k_clip_search builds its candidates in the kernel (integer min/max quantizer, group <= 128, weight-only). The wide
// form takes them from memory: Q [ns][R][K] = the fake-quantized clamped weights of every shrink level, produced by the
// host with the quantizer's own kernels (integer or FP8, per_group / per_channel / per_tensor, v1 clamp or v2 learnable
// range), and the activations twice: XT for the reference output (x * w) and XQT for the candidates' (q_x * q_w) — the
// same buffer when activations are not quantized (auto_clip.py:150,161-166). Both are TRANSPOSED [K][ldt] so that a
// column block lands in LDS by straight 16-byte copies.
// A (row, group) dot product is ATen's vectorized_inner_sum over g elements: 16-element vectors, lo + hi, four
// interleaved streams each a multi_row_sum (16 vectors per stream into level 0, level sums cascaded 16 at a time into
// levels 1 and 2: g <= 262144), the vectors past the last four onto stream 0 AFTER its levels were folded, the trailing
// g % 16 elements first in the final lane sum (oracle/aten_sum.py: inner_sum_16bit, pinned against torch up to
// g = 14336). One wave per row, a lane owns two tokens: its 4 x 8 x 3 x 2 partial sums live in registers while the
// group's columns stream through LDS 128 at a time (the next block's global loads are issued before the current one is
// evaluated); the squared errors of a row go through the same token-sum order as k_clip_search.
template <typename T> __device__ __forceinline__ float bits16_to_f32(uint16_t u) {
    T v;
    v.u = u;
    return to_f32<T>(v);
}
static constexpr int WTOK = 128;      // tokens per pass (2 per lane)
static constexpr int WCOL = 128;      // columns per LDS block
static constexpr int WROWS = 4;       // rows per workgroup (one per wave)

struct ClipCandArgs {
    const void* W;    // [R, K]
    const void* Q;    // [ns, R, K]
    const void* XT;   // [K, ldt]
    const void* XQT;  // [K, ldt]
    int64_t R, K, ldt;
    int g, ng, n_tok, ns;
    void* errs;       // [ns, R, ng]
};

template <typename T>
__global__ __launch_bounds__(256) void k_clip_errs_cand(ClipCandArgs a) {
    constexpr int DT = dt_of<T>::value;
    __shared__ __attribute__((aligned(16))) T xt[WCOL * WTOK];         // [k][token]
    __shared__ __attribute__((aligned(16))) T tab[WROWS * WCOL];       // this block's weights, one row per wave
    __shared__ float sqbuf[WROWS * CTOK];
    __shared__ float esum[WROWS * CMAXS];
    const int gi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row_raw = (int64_t)blockIdx.y * WROWS + wv;
    const bool row_ok = row_raw < a.R;
    const int64_t row = row_ok ? row_raw : a.R - 1;
    const int g = a.g, ns = a.ns;
    const int nblk = (g + WCOL - 1) / WCOL;
    const int nv = g >> 4, nv4 = nv & ~3;
    const bool exact_tok = a.n_tok <= CTOK;
    const bool mode_b = aten_outer_mode_b(a.ng, gi);
    T* mytab = tab + wv * WCOL;
    float* mysq = sqbuf + wv * CTOK;
    const int64_t col0 = (int64_t)gi * g;
    if (tid < WROWS * CMAXS) esum[tid] = 0.f;
    // staging: thread t copies 16 bytes (8 tokens) of column t / 16 + 16 m, m = 0..7
    const int sc = tid >> 4, stok = (tid & 15) * 8;

    for (int t0 = 0; t0 < a.n_tok; t0 += CTOK) {
        const int nt = a.n_tok - t0 < CTOK ? a.n_tok - t0 : CTOK;
        float o0[CTOK / WTOK][2];
        for (int s = 0; s <= ns; ++s) {
            const T* xsrc = (const T*)(s == 0 ? a.XT : a.XQT);
            const T* wsrc = s == 0 ? (const T*)a.W + row * a.K + col0
                                   : (const T*)a.Q + ((int64_t)(s - 1) * a.R + row) * a.K + col0;
            float es_lane = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < CTOK / WTOK; ++pass) {
                const int tb = pass * WTOK;
                if (tb >= nt) break;                                     // uniform
                const int tl = tb + lane * 2;
                float acc[3][4][2][8];
#pragma unroll
                for (int lv = 0; lv < 3; ++lv)
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int l = 0; l < 8; ++l) acc[lv][st][j][l] = 0.f;
                float fin[2] = {0.f, 0.f};
                // prefetch of block 0
                uint4 pre[8];
                uint32_t wpre;
                auto fetch = [&](int b) {
                    const int kb = b * WCOL;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int k = sc + 16 * m;
                        uint4 v = {0u, 0u, 0u, 0u};
                        if (kb + k < g && t0 + tb + stok < a.ldt)
                            v = *(const uint4*)(xsrc + (col0 + kb + k) * a.ldt + t0 + tb + stok);
                        pre[m] = v;
                    }
                    const int k2 = kb + 2 * lane;
                    uint32_t lo = 0u, hi = 0u;
                    if (k2 < g) lo = ((const uint16_t*)wsrc)[k2];
                    if (k2 + 1 < g) hi = ((const uint16_t*)wsrc)[k2 + 1];
                    wpre = lo | (hi << 16);
                };
                fetch(0);
#pragma unroll 1
                for (int b = 0; b < nblk; ++b) {
                    __syncthreads();                                     // block b - 1 fully consumed
#pragma unroll
                    for (int m = 0; m < 8; ++m) *(uint4*)(xt + (sc + 16 * m) * WTOK + stok) = pre[m];
                    ((uint32_t*)mytab)[lane] = wpre;
                    __syncthreads();
                    if (b + 1 < nblk) fetch(b + 1);
                    const int kb = b * WCOL;
                    const int nvb = (g - kb >= WCOL) ? 8 : ((g - kb) >> 4);          // full vectors in this block
                    const int nst = nv4 - 8 * b >= 8 ? 8 : (nv4 - 8 * b > 0 ? nv4 - 8 * b : 0);   // of them on the streams: 0 / 4 / 8
                    auto vec = [&](int i, float (&dst)[2][8]) {
#pragma unroll
                        for (int l = 0; l < 8; ++l) {
                            const int k1 = 16 * i + l, k2 = k1 + 8;
                            const float wa = to_f32<T>(mytab[k1]), wb = to_f32<T>(mytab[k2]);
                            const uint32_t xa = *(const uint32_t*)(xt + k1 * WTOK + lane * 2);
                            const uint32_t xb = *(const uint32_t*)(xt + k2 * WTOK + lane * 2);
                            float p[4];
                            p[0] = bits16_to_f32<T>((uint16_t)(xa & 0xffffu)) * wa;
                            p[1] = bits16_to_f32<T>((uint16_t)(xa >> 16)) * wa;
                            p[2] = bits16_to_f32<T>((uint16_t)(xb & 0xffffu)) * wb;
                            p[3] = bits16_to_f32<T>((uint16_t)(xb >> 16)) * wb;
                            rfast4<DT>(p);
                            dst[0][l] += p[0] + p[2];
                            dst[1][l] += p[1] + p[3];
                        }
                    };
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < nst) vec(i, acc[0][i & 3]);                              // uniform
                    if (nst == 8) {
                        const int rows_done = 2 * (b + 1);                               // vectors per stream so far
                        if ((rows_done & 15) == 0) {
#pragma unroll
                            for (int st = 0; st < 4; ++st)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
#pragma unroll
                                    for (int l = 0; l < 8; ++l) {
                                        acc[1][st][j][l] += acc[0][st][j][l];
                                        acc[0][st][j][l] = 0.f;
                                    }
                            if ((rows_done & 255) == 0) {
#pragma unroll
                                for (int st = 0; st < 4; ++st)
#pragma unroll
                                    for (int j = 0; j < 2; ++j)
#pragma unroll
                                        for (int l = 0; l < 8; ++l) {
                                            acc[2][st][j][l] += acc[1][st][j][l];
                                            acc[1][st][j][l] = 0.f;
                                        }
                            }
                        }
                    }
                    if (b == nblk - 1) {
                        // fold the levels (acc[0] += acc[1]; acc[0] += acc[2]), then the vectors past the last four
                        // onto stream 0, one by one; then the trailing elements, then the lanes
#pragma unroll
                        for (int st = 0; st < 4; ++st)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int l = 0; l < 8; ++l)
                                    acc[0][st][j][l] = (acc[0][st][j][l] + acc[1][st][j][l]) + acc[2][st][j][l];
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (i >= nst && i < nvb) {                                   // at most three
                                float one[2][8];
#pragma unroll
                                for (int j = 0; j < 2; ++j)
#pragma unroll
                                    for (int l = 0; l < 8; ++l) one[j][l] = 0.f;
                                vec(i, one);
#pragma unroll
                                for (int j = 0; j < 2; ++j)
#pragma unroll
                                    for (int l = 0; l < 8; ++l) acc[0][0][j][l] += one[j][l];
                            }
                        for (int k = nvb * 16; k < g - kb; ++k) {
                            const float wk = to_f32<T>(mytab[k]);
                            const uint32_t xk = *(const uint32_t*)(xt + k * WTOK + lane * 2);
                            fin[0] += rfast<DT>(bits16_to_f32<T>((uint16_t)(xk & 0xffffu)) * wk);
                            fin[1] += rfast<DT>(bits16_to_f32<T>((uint16_t)(xk >> 16)) * wk);
                        }
                    }
                }
#pragma unroll
                for (int l = 0; l < 8; ++l)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        fin[j] += ((acc[0][0][j][l] + acc[0][1][j][l]) + acc[0][2][j][l]) + acc[0][3][j][l];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float o = rfast<DT>(fin[j]);
                    if (s == 0) {
                        o0[pass][j] = o;
                    } else {
                        const float d = rfast<DT>(o - o0[pass][j]);
                        const float sq = rfast<DT>(d * d);
                        if (tl + j < nt) {
                            mysq[tl + j] = sq;
                            es_lane += sq;
                        }
                    }
                }
            }
            if (s == 0) continue;
            float tot;
            if (exact_tok) {
                if (!mode_b) {
                    tot = aten_cascade_sum(mysq, 0, 1, nt, lane);
                } else {
                    const int rows4 = nt >> 2;
                    float part[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) part[k4] = aten_cascade_sum(mysq, k4, 4, rows4, lane);
                    for (int i = rows4 * 4; i < nt; ++i) part[0] += mysq[i];
                    tot = ((part[0] + part[1]) + part[2]) + part[3];
                }
                if (lane == 0) esum[wv * CMAXS + s] = tot;
            } else {
                tot = wave_sum(es_lane, 64);
                if (lane == 0) esum[wv * CMAXS + s] += tot;
            }
        }
    }
    if (lane == 0 && row_ok)
        for (int s = 1; s <= ns; ++s)
            ((T*)a.errs)[((int64_t)(s - 1) * a.R + row) * a.ng + gi] =
                from_f32<T>(rndc<DT>(esum[wv * CMAXS + s] / (float)a.n_tok));
}

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_awq_clip_search_ws_bytes(int64_t R, int64_t K, int64_t g, int64_t n_tok) { return 0; }

static int clip_launch(const void* W, const void* X, int dt, int64_t R, int64_t K, int64_t g, int64_t n_tok, int n_grid,
                       int n_shrink, int clip_sym, int sym, float qmin, float qmax, void* best_max, void* best_min,
                       void* errs, llmc_stream_t stream);

extern "C" int llmc_awq_clip_search(const void* W, const void* X, int dt, int64_t R, int64_t K, int64_t g,
                                    int64_t n_tok, int n_grid, int n_shrink, int clip_sym, int sym, float qmin,
                                    float qmax, void* best_max, void* best_min, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(best_max && best_min, "awq_clip_search: null output");
    (void)ws;
    return clip_launch(W, X, dt, R, K, g, n_tok, n_grid, n_shrink, clip_sym, sym, qmin, qmax, best_max, best_min, nullptr,
                       stream);
}

extern "C" int llmc_awq_clip_errs(const void* W, const void* X, int dt, int64_t R, int64_t K, int64_t g, int64_t n_tok,
                                  int n_grid, int n_shrink, int clip_sym, int sym, float qmin, float qmax, void* errs,
                                  llmc_stream_t stream) {
    LLMC_REQUIRE(errs, "awq_clip_errs: null output");
    return clip_launch(W, X, dt, R, K, g, n_tok, n_grid, n_shrink, clip_sym, sym, qmin, qmax, nullptr, nullptr, errs, stream);
}

static int clip_launch(const void* W, const void* X, int dt, int64_t R, int64_t K, int64_t g, int64_t n_tok, int n_grid,
                       int n_shrink, int clip_sym, int sym, float qmin, float qmax, void* best_max, void* best_min,
                       void* errs, llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "awq_clip_search: dtype must be f16 or bf16");
    LLMC_REQUIRE(W && X && R > 0 && K > 0 && n_tok > 0, "awq_clip_search: null/empty argument");
    if (g <= 0) g = K;
    if (g > CG || K % g != 0 || n_shrink < 1 || n_shrink > CMAXS - 1 || n_grid < 1) {
        set_last_error_msg("awq_clip_search: needs group_size <= 128 dividing K and 1..11 shrink steps");
        return LLMC_ENOTSUP;
    }
    hipStream_t st = (hipStream_t)stream;
    ClipArgs a;
    a.W = W; a.X = X; a.R = R; a.K = K; a.g = (int)g; a.ng = (int)(K / g); a.n_tok = (int)n_tok;
    a.n_grid = n_grid; a.n_shrink = n_shrink; a.clip_sym = clip_sym; a.sym = sym; a.qmin = qmin; a.qmax = qmax;
    a.best_max = best_max; a.best_min = best_min; a.errs = errs;
    const size_t lds = (size_t)CG * CTOK * 2 + (size_t)4 * CMAXS * CG * 2 + (size_t)(CROWS * CMAXS + 4 * CTOK) * sizeof(float);
    dim3 grid((unsigned)a.ng, (unsigned)ceil_div64(R, CROWS));
    if (dt == LLMC_F16) {
        if (int rc = ensure_dynamic_lds((const void*)k_clip_search<f16_t>, (int)lds)) return rc;
        hipLaunchKernelGGL((k_clip_search<f16_t>), grid, dim3(256), lds, st, a);
    } else {
        if (int rc = ensure_dynamic_lds((const void*)k_clip_search<bf16_t>, (int)lds)) return rc;
        hipLaunchKernelGGL((k_clip_search<bf16_t>), grid, dim3(256), lds, st, a);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_awq_clip_errs_cand(const void* W, const void* Q, const void* XT, const void* XQT, int dt, int64_t R,
                                       int64_t K, int64_t g, int64_t n_tok, int64_t ldt, int n_shrink, void* errs,
                                       llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "awq_clip_errs_cand: dtype must be f16 or bf16");
    LLMC_REQUIRE(W && Q && XT && XQT && errs && R > 0 && K > 0 && n_tok > 0, "awq_clip_errs_cand: null/empty argument");
    if (g <= 0) g = K;
    LLMC_REQUIRE(K % g == 0 && g >= 16 && g <= 262144, "awq_clip_errs_cand: group size must divide K, 16 <= g <= 262144");
    LLMC_REQUIRE(n_shrink >= 1 && n_shrink <= CMAXS - 1, "awq_clip_errs_cand: 1..11 shrink steps");
    LLMC_REQUIRE(ldt >= n_tok && ldt % 8 == 0, "awq_clip_errs_cand: ldt must be a multiple of 8 tokens >= n_tok");
    LLMC_REQUIRE(((uintptr_t)XT | (uintptr_t)XQT) % 16 == 0, "awq_clip_errs_cand: XT / XQT must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ClipCandArgs a;
    a.W = W; a.Q = Q; a.XT = XT; a.XQT = XQT; a.R = R; a.K = K; a.ldt = ldt; a.g = (int)g; a.ng = (int)(K / g);
    a.n_tok = (int)n_tok; a.ns = n_shrink; a.errs = errs;
    dim3 grid((unsigned)a.ng, (unsigned)ceil_div64(R, WROWS));
    if (dt == LLMC_F16) hipLaunchKernelGGL((k_clip_errs_cand<f16_t>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_clip_errs_cand<bf16_t>), grid, dim3(256), 0, st, a);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
