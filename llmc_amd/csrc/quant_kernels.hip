// quant_kernels.hip — HBM-bound quantizer kernels (K5/K6/K7 of SURVEY.md §2.3):
//   min/max -> scale/zero, round/clamp (fake + real), LSB-first int packing.
// Layout: W is a contiguous [G, g] view (one quantization group per row). A wave64 is cut into
// 64/LPR sub-groups of LPR lanes, one row per sub-group, 16 B per lane per load (coalesced: a sub-group
// reads LPR*16 contiguous bytes). Rows are distributed over a grid-stride of waves.
#include "common.h"
#include "quant_math.h"

namespace llmc {

static constexpr int kBlock = 256;
static constexpr int kMaxGrid = 256 * 8;

template <typename T, int VEC> struct RowVec {
    T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ RowVec<T, VEC> load_vec(const T* p) {
    RowVec<T, VEC> r;
    if constexpr (VEC * sizeof(T) == 16) {
        uint4 raw = *reinterpret_cast<const uint4*>(p);
        __builtin_memcpy(&r, &raw, 16);
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) r.v[i] = p[i];
    }
    return r;
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const RowVec<T, VEC>& r) {
    if constexpr (VEC * sizeof(T) == 16) {
        uint4 raw;
        __builtin_memcpy(&raw, &r, 16);
        *reinterpret_cast<uint4*>(p) = raw;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) p[i] = r.v[i];
    }
}

template <int KIND> struct code_t;
template <> struct code_t<LLMC_OUT_I32> { using type = int32_t; };
template <> struct code_t<LLMC_OUT_I8> { using type = int8_t; };
template <> struct code_t<LLMC_OUT_U8> { using type = uint8_t; };

// --------------------------------------------------------------------------------------------
// row scan: min/max of one row by LPR lanes
// --------------------------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void row_minmax(const T* row, int g, int sl, int lpr, float& mn, float& mx,
                                           RowVec<T, VEC>& first, bool& have_first) {
    mn = INFINITY;
    mx = -INFINITY;
    have_first = false;
    for (int c = sl * VEC; c < g; c += lpr * VEC) {
        RowVec<T, VEC> v = load_vec<T, VEC>(row + c);
        if (!have_first) {
            first = v;
            have_first = true;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float f = to_f32<T>(v.v[i]);
            mn = fminf(mn, f);
            mx = fmaxf(mx, f);
        }
    }
    mn = wave_min(mn, lpr);
    mx = wave_max(mx, lpr);
}

// K5: [G, g] -> scales/zeros [G] (tensor dtype)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_minmax_qparams(const T* __restrict__ W, int64_t G, int g,
                                                           int lpr, int sym, int round_zp, float qmin,
                                                           float qmax, T* __restrict__ scales,
                                                           T* __restrict__ zeros) {
    constexpr int DT = dt_of<T>::value;
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const int sub = lane / lpr, sl = lane % lpr;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t r0 = wave * rpw; r0 < G; r0 += nwaves * rpw) {
        int64_t row = r0 + sub;
        bool valid = row < G;
        int64_t rr = valid ? row : G - 1;
        float mn, mx;
        RowVec<T, VEC> first;
        bool hf;
        row_minmax<T, VEC>(W + rr * g, g, sl, lpr, mn, mx, first, hf);
        if (valid && sl == 0) {
            QParams q = qparams_from_minmax(mn, mx, DT, sym, round_zp, qmin, qmax);
            scales[row] = from_f32<T>(q.s);
            if (zeros) zeros[row] = from_f32<T>(q.z);
        }
    }
}

// two-stage variant for few, very long rows (per_tensor / huge per_channel)
static constexpr int kChunk = 8192;
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_minmax_partial(const T* __restrict__ W, int64_t G, int64_t g,
                                                           int64_t nch, float2* __restrict__ part) {
    __shared__ float smn[kBlock / 64], smx[kBlock / 64];
    for (int64_t u = blockIdx.x; u < G * nch; u += gridDim.x) {
        int64_t row = u / nch, ch = u % nch;
        int64_t c0 = ch * kChunk, c1 = c0 + kChunk < g ? c0 + kChunk : g;
        const T* p = W + row * g;
        float mn = INFINITY, mx = -INFINITY;
        for (int64_t c = c0 + (int64_t)threadIdx.x * VEC; c < c1; c += (int64_t)kBlock * VEC) {
            RowVec<T, VEC> v = load_vec<T, VEC>(p + c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float f = to_f32<T>(v.v[i]);
                mn = fminf(mn, f);
                mx = fmaxf(mx, f);
            }
        }
        mn = wave_min(mn, 64);
        mx = wave_max(mx, 64);
        if ((threadIdx.x & 63) == 0) {
            smn[threadIdx.x >> 6] = mn;
            smx[threadIdx.x >> 6] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < kBlock / 64; ++i) {
                mn = fminf(mn, smn[i]);
                mx = fmaxf(mx, smx[i]);
            }
            part[u] = make_float2(mn, mx);
        }
        __syncthreads();
    }
}
template <typename T>
__global__ void k_minmax_final(const float2* __restrict__ part, int64_t G, int64_t nch, int sym,
                               int round_zp, float qmin, float qmax, T* __restrict__ scales,
                               T* __restrict__ zeros) {
    constexpr int DT = dt_of<T>::value;
    __shared__ float smn[16], smx[16];
    int64_t row = blockIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t c = threadIdx.x; c < nch; c += 1024) {
        float2 p = part[row * nch + c];
        mn = fminf(mn, p.x);
        mx = fmaxf(mx, p.y);
    }
    mn = wave_min(mn, 64);
    mx = wave_max(mx, 64);
    if ((threadIdx.x & 63) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) {
            mn = fminf(mn, smn[i]);
            mx = fmaxf(mx, smx[i]);
        }
        QParams q = qparams_from_minmax(mn, mx, DT, sym, round_zp, qmin, qmax);
        scales[row] = from_f32<T>(q.s);
        if (zeros) zeros[row] = from_f32<T>(q.z);
    }
}

// K6 static: given qparams
template <typename T, int VEC, int KIND>
__global__ __launch_bounds__(kBlock) void k_quant_static(const T* __restrict__ W, int64_t G, int g,
                                                         const void* __restrict__ scales, int sdt,
                                                         const void* __restrict__ zeros, int zdt,
                                                         float qmin, float qmax, void* __restrict__ out) {
    constexpr int WDT = dt_of<T>::value;
    // LLMC_SCALAR_QPARAM: a 0-dim operand keeps its own precision but does not take part in type promotion
    const int sd = sdt & 3, zd = zdt & 3;
    const bool fz = (zdt & LLMC_FRACTIONAL_ZP) != 0;
    const int p1 = (sdt & LLMC_SCALAR_QPARAM) ? WDT : promote(WDT, sd);
    const int p2 = (zeros && !(zdt & LLMC_SCALAR_QPARAM)) ? promote(p1, zd) : p1;
    const int64_t nvec_row = g / VEC;
    const int64_t total = G * nvec_row;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kBlock) {
        int64_t row = i / nvec_row;
        int64_t c = (i - row * nvec_row) * VEC;
        float s = load_as_f32(scales, row, sd);
        float z = zeros ? load_as_f32(zeros, row, zd) : 0.0f;
        RowVec<T, VEC> v = load_vec<T, VEC>(W + row * g + c);
        float am = 0.0f;   // bound of |x| over this thread's elements for the hoisted divisor (quant_math.h)
#pragma unroll
        for (int k = 0; k < VEC; ++k) am = fmaxf(am, fabsf(to_f32<T>(v.v[k])));
        // LLMC_FRACTIONAL_ZP (round_zp=False, quant.py:702-707): the divisor is s.clamp_min(1e-9) in the scale's dtype
        const float sdiv = fz ? fmaxf(s, rnd(1e-9f, sd)) : s;
        const Divisor dv = make_divisor(sdiv, am);
        if constexpr (KIND == LLMC_OUT_FAKE) {
            RowVec<T, VEC> o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float q = fz ? quant_code_fz(to_f32<T>(v.v[k]), dv, z, p1, p2, qmin, qmax)
                             : quant_code(to_f32<T>(v.v[k]), dv, z, p1, p2, qmin, qmax);
                o.v[k] = from_f32<T>(dequant_code(q, s, z, p2));
            }
            store_vec<T, VEC>((T*)out + row * g + c, o);
        } else {
            using C = typename code_t<KIND>::type;
            RowVec<C, VEC> o;
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                o.v[k] = (C)(fz ? quant_code_fz(to_f32<T>(v.v[k]), dv, z, p1, p2, qmin, qmax)
                                : quant_code(to_f32<T>(v.v[k]), dv, z, p1, p2, qmin, qmax));
            C* op = (C*)out + row * g + c;
#pragma unroll
            for (int k = 0; k < VEC; ++k) op[k] = o.v[k];
        }
    }
}

// K5+K6 fused dynamic: one pass over W from HBM (second row pass hits L1/L2; the first 16 B per lane
// stay in registers, which covers g <= LPR*VEC, i.e. every per_group case).
template <typename T, int VEC, int KIND>
__global__ __launch_bounds__(kBlock) void k_quant_dynamic(const T* __restrict__ W, int64_t G, int g,
                                                          int lpr, int sym, int round_zp, float qmin,
                                                          float qmax, void* __restrict__ out,
                                                          T* __restrict__ scales, T* __restrict__ zeros) {
    constexpr int DT = dt_of<T>::value;
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const int sub = lane / lpr, sl = lane % lpr;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t r0 = wave * rpw; r0 < G; r0 += nwaves * rpw) {
        int64_t row = r0 + sub;
        bool valid = row < G;
        int64_t rr = valid ? row : G - 1;
        const T* rp = W + rr * g;
        float mn, mx;
        RowVec<T, VEC> first;
        bool hf;
        row_minmax<T, VEC>(rp, g, sl, lpr, mn, mx, first, hf);
        QParams q = qparams_from_minmax(mn, mx, DT, sym, round_zp, qmin, qmax);
        if (valid && sl == 0) {
            if (scales) scales[row] = from_f32<T>(q.s);
            if (zeros) zeros[row] = from_f32<T>(q.z);
        }
        if (!valid) continue;
        const Divisor dv = make_divisor(q.s, fmaxf(fabsf(mn), fabsf(mx)));
        bool use_first = true;
        for (int c = sl * VEC; c < g; c += lpr * VEC) {
            RowVec<T, VEC> v = use_first ? first : load_vec<T, VEC>(rp + c);
            use_first = false;
            if constexpr (KIND == LLMC_OUT_FAKE) {
                RowVec<T, VEC> o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float qq = quant_code(to_f32<T>(v.v[k]), dv, q.z, DT, DT, qmin, qmax);
                    o.v[k] = from_f32<T>(dequant_code(qq, q.s, q.z, DT));
                }
                store_vec<T, VEC>((T*)out + rr * g + c, o);
            } else {
                using C = typename code_t<KIND>::type;
                C* op = (C*)out + rr * g + c;
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    op[k] = (C)quant_code(to_f32<T>(v.v[k]), dv, q.z, DT, DT, qmin, qmax);
            }
        }
    }
}

// ---- fast path for short rows (per_group: g == LPR * VEC exactly, one 16-B vector per lane per row):
// UNR independent row-sets per wave iteration keep 4 loads per lane in flight (the generic loop has one and
// measured 2.2-2.9 TB/s; HBM latency x bandwidth needs ~12 KB in flight per SIMD).
static constexpr int UNR = 4;
template <typename T, int VEC, int KIND, bool SCALE>
__global__ __launch_bounds__(kBlock) void k_quant_dynamic_small(const T* __restrict__ W, const T* __restrict__ cs,
                                                                int64_t G, int g, int gpr, int lpr, int sym,
                                                                int round_zp, float qmin, float qmax,
                                                                void* __restrict__ out, T* __restrict__ scales,
                                                                T* __restrict__ zeros) {
    constexpr int DT = dt_of<T>::value;
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const int sub = lane / lpr, sl = lane % lpr;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t r0 = wave * rpw * UNR; r0 < G; r0 += nwaves * rpw * UNR) {
        RowVec<T, VEC> v[UNR];
        bool valid[UNR];
        int64_t rows[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t row = r0 + u * rpw + sub;
            valid[u] = row < G;
            rows[u] = valid[u] ? row : G - 1;
            v[u] = load_vec<T, VEC>(W + rows[u] * g + sl * VEC);
        }
        if (SCALE) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                RowVec<T, VEC> sv = load_vec<T, VEC>(cs + (rows[u] % gpr) * g + sl * VEC);
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    v[u].v[i] = from_f32<T>(rndc<DT>(to_f32<T>(v[u].v[i]) * to_f32<T>(sv.v[i])));
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float mn = INFINITY, mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float f = to_f32<T>(v[u].v[i]);
                mn = fminf(mn, f);
                mx = fmaxf(mx, f);
            }
            mn = wave_min(mn, lpr);
            mx = wave_max(mx, lpr);
            const QParams q = qparams_from_minmax(mn, mx, DT, sym, round_zp, qmin, qmax);
            if (valid[u] && sl == 0) {
                if (scales) scales[rows[u]] = from_f32<T>(q.s);
                if (zeros) zeros[rows[u]] = from_f32<T>(q.z);
            }
            if (!valid[u] || out == nullptr) continue;
            const Divisor dv = make_divisor(q.s, fmaxf(fabsf(mn), fabsf(mx)));
            if constexpr (KIND == LLMC_OUT_FAKE) {
                RowVec<T, VEC> o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float qq = quant_code(to_f32<T>(v[u].v[k]), dv, q.z, DT, DT, qmin, qmax);
                    o.v[k] = from_f32<T>(dequant_code(qq, q.s, q.z, DT));
                }
                store_vec<T, VEC>((T*)out + rows[u] * g + sl * VEC, o);
            } else {
                using C = typename code_t<KIND>::type;
                RowVec<C, VEC> o;
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    o.v[k] = (C)quant_code(to_f32<T>(v[u].v[k]), dv, q.z, DT, DT, qmin, qmax);
                C* op = (C*)out + rows[u] * g + sl * VEC;
                if constexpr (sizeof(C) * VEC == 32) {
                    uint4 lo, hi;
                    __builtin_memcpy(&lo, &o.v[0], 16);
                    __builtin_memcpy(&hi, &o.v[VEC / 2], 16);
                    reinterpret_cast<uint4*>(op)[0] = lo;
                    reinterpret_cast<uint4*>(op)[1] = hi;
                } else if constexpr (sizeof(C) * VEC == 16) {
                    uint4 lo;
                    __builtin_memcpy(&lo, &o.v[0], 16);
                    reinterpret_cast<uint4*>(op)[0] = lo;
                } else if constexpr (sizeof(C) * VEC == 8) {
                    uint2 lo;
                    __builtin_memcpy(&lo, &o.v[0], 8);
                    reinterpret_cast<uint2*>(op)[0] = lo;
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) op[k] = o.v[k];
                }
            }
        }
    }
}

static inline bool small_ok(int64_t g, int vec) {
    int64_t lpr = g / vec;
    return g % vec == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
}

// K7: LSB-first packing. One thread per output word; 32/bits consecutive codes -> one int32.
template <typename C, bool VECOK>
__global__ __launch_bounds__(kBlock) void k_pack_lsb(const C* __restrict__ codes, int64_t R, int64_t K,
                                                     int bits, int64_t Kp, int32_t* __restrict__ packed) {
    const int pf = 32 / bits;
    const int off = 1 << (bits - 1);
    const int64_t total = R * Kp;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kBlock) {
        int64_t r = i / Kp, j = i - r * Kp;
        const C* p = codes + r * K + j * pf;
        uint32_t w = 0;
        if (j * pf + pf <= K) {
            if constexpr (sizeof(C) == 4 && VECOK) {
                if (pf == 8) {
                    int4 a = *reinterpret_cast<const int4*>(p);
                    int4 b = *reinterpret_cast<const int4*>(p + 4);
                    int v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) w |= (uint32_t)((v[k] + off) & 0xff) << (bits * k);
                } else {
                    for (int k = 0; k < pf; ++k) w |= (uint32_t)(((int)p[k] + off) & 0xff) << (bits * k);
                }
            } else {
                for (int k = 0; k < pf; ++k) w |= (uint32_t)(((int)p[k] + off) & 0xff) << (bits * k);
            }
        } else {
            for (int k = 0; k < pf && j * pf + k < K; ++k)
                w |= (uint32_t)(((int)p[k] + off) & 0xff) << (bits * k);
        }
        packed[i] = (int32_t)w;
    }
}

static inline int grid_for(int64_t work_items, int per_block) {
    int64_t b = ceil_div64(work_items, per_block);
    if (b < 1) b = 1;
    return (int)(b > kMaxGrid ? kMaxGrid : b);
}

static inline int choose_lpr(int64_t g, int vec) {
    int lpr = pow2_ceil(ceil_div64(g, vec));
    if (lpr > 64) lpr = 64;
    if (lpr < 1) lpr = 1;
    return lpr;
}

static inline bool use_two_stage(int64_t G, int64_t g) { return g >= 4 * kChunk && G < 4096; }

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_minmax_qparams_ws_bytes(int64_t G, int64_t g) {
    if (G <= 0 || g <= 0) return 0;
    if (!use_two_stage(G, g)) return 0;
    return (size_t)(G * ceil_div64(g, kChunk)) * sizeof(float2);
}

template <typename T>
static int minmax_qparams_t(const void* W, int64_t G, int64_t g, int sym, int round_zp, float qmin,
                            float qmax, void* scales, void* zeros, void* ws, hipStream_t st) {
    constexpr int V16 = 16 / sizeof(T);
    bool vec_ok = (g % V16 == 0) && (((uintptr_t)W & 15) == 0);
    if (use_two_stage(G, g)) {
        LLMC_REQUIRE(ws != nullptr, "minmax_qparams: workspace required for long rows");
        int64_t nch = ceil_div64(g, kChunk);
        int grid = grid_for(G * nch, 1);
        if (vec_ok)
            hipLaunchKernelGGL((k_minmax_partial<T, V16>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G, g,
                               nch, (float2*)ws);
        else
            hipLaunchKernelGGL((k_minmax_partial<T, 1>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G, g,
                               nch, (float2*)ws);
        LLMC_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_minmax_final<T>), dim3((unsigned)G), dim3(1024), 0, st, (const float2*)ws, G, nch,
                           sym, round_zp, qmin, qmax, (T*)scales, (T*)zeros);
        LLMC_LAUNCH_CHECK();
        return LLMC_OK;
    }
    LLMC_REQUIRE(g < (1ll << 31), "minmax_qparams: row too long");
    if (vec_ok && small_ok(g, V16)) {
        int lpr = (int)(g / V16);
        int grid = grid_for(ceil_div64(G, (64 / lpr) * UNR), kBlock / 64);
        hipLaunchKernelGGL((k_quant_dynamic_small<T, V16, LLMC_OUT_FAKE, false>), dim3(grid), dim3(kBlock), 0, st,
                           (const T*)W, (const T*)nullptr, G, (int)g, 1, lpr, sym, round_zp, qmin, qmax, (void*)nullptr,
                           (T*)scales, (T*)zeros);
        LLMC_LAUNCH_CHECK();
        return LLMC_OK;
    }
    if (vec_ok) {
        int lpr = choose_lpr(g, V16);
        int grid = grid_for(ceil_div64(G, 64 / lpr), kBlock / 64);
        hipLaunchKernelGGL((k_minmax_qparams<T, V16>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G, (int)g,
                           lpr, sym, round_zp, qmin, qmax, (T*)scales, (T*)zeros);
    } else {
        int lpr = choose_lpr(g, 1);
        int grid = grid_for(ceil_div64(G, 64 / lpr), kBlock / 64);
        hipLaunchKernelGGL((k_minmax_qparams<T, 1>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G, (int)g,
                           lpr, sym, round_zp, qmin, qmax, (T*)scales, (T*)zeros);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_minmax_qparams(const void* W, int dt, int64_t G, int64_t g, int sym, int round_zp,
                                   float qmin, float qmax, void* scales, void* zeros, void* ws,
                                   llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt), "minmax_qparams: bad dtype");
    LLMC_REQUIRE(W && scales && G > 0 && g > 0, "minmax_qparams: null/empty argument");
    LLMC_REQUIRE(sym || zeros, "minmax_qparams: zeros required for asymmetric");
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case LLMC_F16: return minmax_qparams_t<f16_t>(W, G, g, sym, round_zp, qmin, qmax, scales, zeros, ws, st);
        case LLMC_BF16: return minmax_qparams_t<bf16_t>(W, G, g, sym, round_zp, qmin, qmax, scales, zeros, ws, st);
        default: return minmax_qparams_t<float>(W, G, g, sym, round_zp, qmin, qmax, scales, zeros, ws, st);
    }
}


// calib_algo = 'mse' (BaseQuantizer.get_mse_range, quant.py:145-203) + get_qparams on the searched range.
// One wave per row of the [G, g] view. The reference works on tensor.float(): ranges, qparams and the fake-quant are
// fp32. Its candidate ranges COMPOUND: best_min_val aliases _min_val, so after an improvement at step i the next
// candidate is p_{i+1} times the already shrunk range (oracle/quant_ref.py:mse_range pins this against the goldens).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mse_qparams(const T* __restrict__ W, int64_t G, int g, int sym,
                                                        int round_zp, float qmin, float qmax, int nsteps, int grid,
                                                        float norm, float* __restrict__ scales,
                                                        float* __restrict__ zeros, float* __restrict__ min_out,
                                                        float* __restrict__ max_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t row = wave; row < G; row += nwaves) {
        const T* w = W + row * g;
        float mn = INFINITY, mx = -INFINITY;
        for (int c = lane; c < g; c += 64) {
            const float x = to_f32<T>(w[c]);
            mn = fminf(mn, x);
            mx = fmaxf(mx, x);
        }
        float cur_min = wave_min(mn, 64), cur_max = wave_max(mx, 64);
        const float row_absmax = fmaxf(fabsf(cur_min), fabsf(cur_max));
        float best = INFINITY;
        for (int i = 0; i < nsteps; ++i) {
            const float p = (float)(1.0 - (double)i / (double)grid);   // python float -> fp32 scalar operand
            const float xmin = p * cur_min, xmax = p * cur_max;
            const QParams q = qparams_from_minmax(xmin, xmax, LLMC_F32, sym, round_zp, qmin, qmax);
            const Divisor dv = make_divisor(q.s, row_absmax);
            float acc = 0.0f;
            for (int c = lane; c < g; c += 64) {
                const float x = to_f32<T>(w[c]);
                const float code = quant_code(x, dv, q.z, LLMC_F32, LLMC_F32, qmin, qmax);
                const float d = fabsf(dequant_code(code, q.s, q.z, LLMC_F32) - x);
                acc += powf(d, norm);
            }
            const float err = wave_sum(acc, 64);
            if (err < best) {
                best = err;
                cur_min = xmin;
                cur_max = xmax;
            }
        }
        if (lane == 0) {
            const QParams q = qparams_from_minmax(cur_min, cur_max, LLMC_F32, sym, round_zp, qmin, qmax);
            scales[row] = q.s;
            if (zeros) zeros[row] = q.z;
            if (min_out) min_out[row] = cur_min;
            if (max_out) max_out[row] = cur_max;
        }
    }
}

extern "C" int llmc_mse_qparams(const void* W, int dt, int64_t G, int64_t g, int sym, int round_zp, float qmin,
                                float qmax, int nsteps, int grid, float norm, float* scales, float* zeros,
                                float* min_out, float* max_out, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt), "mse_qparams: bad dtype");
    LLMC_REQUIRE(W && scales && G > 0 && g > 0 && g < (1ll << 31), "mse_qparams: null/empty argument");
    LLMC_REQUIRE(sym || zeros, "mse_qparams: zeros required for asymmetric");
    LLMC_REQUIRE(nsteps >= 1 && grid >= 1, "mse_qparams: nsteps and grid must be positive");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = grid_for(G, kBlock / 64);
    switch (dt) {
        case LLMC_F16:
            hipLaunchKernelGGL((k_mse_qparams<f16_t>), dim3(nblk), dim3(kBlock), 0, st, (const f16_t*)W, G, (int)g, sym,
                               round_zp, qmin, qmax, nsteps, grid, norm, scales, zeros, min_out, max_out);
            break;
        case LLMC_BF16:
            hipLaunchKernelGGL((k_mse_qparams<bf16_t>), dim3(nblk), dim3(kBlock), 0, st, (const bf16_t*)W, G, (int)g, sym,
                               round_zp, qmin, qmax, nsteps, grid, norm, scales, zeros, min_out, max_out);
            break;
        default:
            hipLaunchKernelGGL((k_mse_qparams<float>), dim3(nblk), dim3(kBlock), 0, st, (const float*)W, G, (int)g, sym,
                               round_zp, qmin, qmax, nsteps, grid, norm, scales, zeros, min_out, max_out);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

template <typename T, int KIND>
static int quant_static_tk(const void* W, int64_t G, int64_t g, const void* scales, int sdt,
                           const void* zeros, int zdt, float qmin, float qmax, void* out, hipStream_t st) {
    constexpr int V16 = 16 / sizeof(T);
    bool vec_ok = (g % V16 == 0) && (((uintptr_t)W & 15) == 0) &&
                  (KIND != LLMC_OUT_FAKE || ((uintptr_t)out & 15) == 0);
    LLMC_REQUIRE(g < (1ll << 31), "quant_static: row too long");
    if (vec_ok) {
        int grid = grid_for(G * (g / V16), kBlock);
        hipLaunchKernelGGL((k_quant_static<T, V16, KIND>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G,
                           (int)g, scales, sdt, zeros, zdt, qmin, qmax, out);
    } else {
        int grid = grid_for(G * g, kBlock);
        hipLaunchKernelGGL((k_quant_static<T, 1, KIND>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G,
                           (int)g, scales, sdt, zeros, zdt, qmin, qmax, out);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
template <typename T>
static int quant_static_t(const void* W, int64_t G, int64_t g, const void* scales, int sdt,
                          const void* zeros, int zdt, float qmin, float qmax, int kind, void* out,
                          hipStream_t st) {
    switch (kind) {
        case LLMC_OUT_FAKE: return quant_static_tk<T, LLMC_OUT_FAKE>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out, st);
        case LLMC_OUT_I32: return quant_static_tk<T, LLMC_OUT_I32>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out, st);
        case LLMC_OUT_I8: return quant_static_tk<T, LLMC_OUT_I8>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out, st);
        case LLMC_OUT_U8: return quant_static_tk<T, LLMC_OUT_U8>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out, st);
    }
    set_last_error_msg("quant_static: bad out_kind");
    return LLMC_EINVAL;
}

extern "C" int llmc_quant_static(const void* W, int wdt, int64_t G, int64_t g, const void* scales, int sdt,
                                 const void* zeros, int zdt, float qmin, float qmax, int out_kind,
                                 void* out, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(wdt) && dtype_ok(sdt & ~LLMC_SCALAR_QPARAM) && (!zeros || dtype_ok(zdt & ~(LLMC_SCALAR_QPARAM | LLMC_FRACTIONAL_ZP))),
                 "quant_static: bad dtype");
    LLMC_REQUIRE(W && scales && out && G > 0 && g > 0, "quant_static: null/empty argument");
    hipStream_t st = (hipStream_t)stream;
    switch (wdt) {
        case LLMC_F16: return quant_static_t<f16_t>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out_kind, out, st);
        case LLMC_BF16: return quant_static_t<bf16_t>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out_kind, out, st);
        default: return quant_static_t<float>(W, G, g, scales, sdt, zeros, zdt, qmin, qmax, out_kind, out, st);
    }
}

extern "C" size_t llmc_quant_dynamic_ws_bytes(int64_t G, int64_t g) {
    if (G <= 0 || g <= 0) return 0;
    if (!use_two_stage(G, g)) return 0;
    // partial min/max + a private copy of scales/zeros when the caller does not want them
    return llmc_minmax_qparams_ws_bytes(G, g) + (size_t)G * 8 + 64;
}

template <typename T, int KIND>
static int quant_dynamic_tk(const void* W, int64_t G, int64_t g, int sym, int round_zp, float qmin,
                            float qmax, void* out, void* scales, void* zeros, hipStream_t st) {
    constexpr int V16 = 16 / sizeof(T);
    bool vec_ok = (g % V16 == 0) && (((uintptr_t)W & 15) == 0) &&
                  (KIND != LLMC_OUT_FAKE || ((uintptr_t)out & 15) == 0);
    LLMC_REQUIRE(g < (1ll << 31), "quant_dynamic: row too long");
    if (vec_ok && small_ok(g, V16) && (((uintptr_t)out & 15) == 0)) {
        int lpr = (int)(g / V16);
        int grid = grid_for(ceil_div64(G, (64 / lpr) * UNR), kBlock / 64);
        hipLaunchKernelGGL((k_quant_dynamic_small<T, V16, KIND, false>), dim3(grid), dim3(kBlock), 0, st, (const T*)W,
                           (const T*)nullptr, G, (int)g, 1, lpr, sym, round_zp, qmin, qmax, out, (T*)scales, (T*)zeros);
        LLMC_LAUNCH_CHECK();
        return LLMC_OK;
    }
    if (vec_ok) {
        int lpr = choose_lpr(g, V16);
        int grid = grid_for(ceil_div64(G, 64 / lpr), kBlock / 64);
        hipLaunchKernelGGL((k_quant_dynamic<T, V16, KIND>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G,
                           (int)g, lpr, sym, round_zp, qmin, qmax, out, (T*)scales, (T*)zeros);
    } else {
        int lpr = choose_lpr(g, 1);
        int grid = grid_for(ceil_div64(G, 64 / lpr), kBlock / 64);
        hipLaunchKernelGGL((k_quant_dynamic<T, 1, KIND>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, G,
                           (int)g, lpr, sym, round_zp, qmin, qmax, out, (T*)scales, (T*)zeros);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
template <typename T>
static int quant_dynamic_t(const void* W, int64_t G, int64_t g, int sym, int round_zp, float qmin,
                           float qmax, int kind, void* out, void* scales, void* zeros, hipStream_t st) {
    switch (kind) {
        case LLMC_OUT_FAKE: return quant_dynamic_tk<T, LLMC_OUT_FAKE>(W, G, g, sym, round_zp, qmin, qmax, out, scales, zeros, st);
        case LLMC_OUT_I32: return quant_dynamic_tk<T, LLMC_OUT_I32>(W, G, g, sym, round_zp, qmin, qmax, out, scales, zeros, st);
        case LLMC_OUT_I8: return quant_dynamic_tk<T, LLMC_OUT_I8>(W, G, g, sym, round_zp, qmin, qmax, out, scales, zeros, st);
        case LLMC_OUT_U8: return quant_dynamic_tk<T, LLMC_OUT_U8>(W, G, g, sym, round_zp, qmin, qmax, out, scales, zeros, st);
    }
    set_last_error_msg("quant_dynamic: bad out_kind");
    return LLMC_EINVAL;
}

extern "C" int llmc_quant_dynamic(const void* W, int dt, int64_t G, int64_t g, int sym, int round_zp,
                                  float qmin, float qmax, int out_kind, void* out, void* scales_out,
                                  void* zeros_out, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt), "quant_dynamic: bad dtype");
    LLMC_REQUIRE(W && out && G > 0 && g > 0, "quant_dynamic: null/empty argument");
    LLMC_REQUIRE(round_zp == 1, "quant_dynamic: only round_zp=True is supported");
    hipStream_t st = (hipStream_t)stream;
    if (use_two_stage(G, g)) {
        // few long rows (per_tensor): min/max by the two-stage reduction, then the static kernel
        LLMC_REQUIRE(ws != nullptr, "quant_dynamic: workspace required for long rows");
        char* wsb = (char*)ws;
        size_t off = (llmc_minmax_qparams_ws_bytes(G, g) + 63) & ~(size_t)63;
        void* s = scales_out ? scales_out : (void*)(wsb + off);
        void* z = sym ? nullptr : (zeros_out ? zeros_out : (void*)(wsb + off + (size_t)G * 4));
        int rc = llmc_minmax_qparams(W, dt, G, g, sym, round_zp, qmin, qmax, s, sym ? zeros_out : z, ws, stream);
        if (rc) return rc;
        return llmc_quant_static(W, dt, G, g, s, dt, z, dt, qmin, qmax, out_kind, out, stream);
    }
    void* zo = sym ? nullptr : zeros_out;
    int rc;
    switch (dt) {
        case LLMC_F16: rc = quant_dynamic_t<f16_t>(W, G, g, sym, round_zp, qmin, qmax, out_kind, out, scales_out, zo, st); break;
        case LLMC_BF16: rc = quant_dynamic_t<bf16_t>(W, G, g, sym, round_zp, qmin, qmax, out_kind, out, scales_out, zo, st); break;
        default: rc = quant_dynamic_t<float>(W, G, g, sym, round_zp, qmin, qmax, out_kind, out, scales_out, zo, st); break;
    }
    if (rc) return rc;
    if (sym && zeros_out) LLMC_HIP_CHECK(hipMemsetAsync(zeros_out, 0, (size_t)G * dtype_size(dt), st));
    return LLMC_OK;
}

// fake_quantize_weight of AWQ's search (awq.py:147-164): w' = rnd(w * s[col]) (the in-place mul_ in the
// model dtype), then the dynamic fake-quant of the scaled row group. One pass over W.
namespace llmc {
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_scale_fakequant(const T* __restrict__ W, const T* __restrict__ cs,
                                                            int64_t G, int g, int gpr /*groups per row*/,
                                                            int lpr, int sym, float qmin, float qmax,
                                                            T* __restrict__ out) {
    constexpr int DT = dt_of<T>::value;
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const int sub = lane / lpr, sl = lane % lpr;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t r0 = wave * rpw; r0 < G; r0 += nwaves * rpw) {
        int64_t row = r0 + sub;
        const bool valid = row < G;
        const int64_t rr = valid ? row : G - 1;
        const T* rp = W + rr * g;
        const T* cp = cs + (rr % gpr) * g;
        float mn = INFINITY, mx = -INFINITY;
        RowVec<T, VEC> first;
        bool hf = false;
        for (int c = sl * VEC; c < g; c += lpr * VEC) {
            RowVec<T, VEC> v = load_vec<T, VEC>(rp + c);
            RowVec<T, VEC> sv = load_vec<T, VEC>(cp + c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float f = rndc<DT>(to_f32<T>(v.v[i]) * to_f32<T>(sv.v[i]));
                v.v[i] = from_f32<T>(f);
                mn = fminf(mn, f);
                mx = fmaxf(mx, f);
            }
            if (!hf) {
                first = v;
                hf = true;
            }
        }
        mn = wave_min(mn, lpr);
        mx = wave_max(mx, lpr);
        QParams q = qparams_from_minmax(mn, mx, DT, sym, 1, qmin, qmax);
        if (!valid) continue;
        const Divisor dv = make_divisor(q.s, fmaxf(fabsf(mn), fabsf(mx)));
        bool use_first = true;
        for (int c = sl * VEC; c < g; c += lpr * VEC) {
            RowVec<T, VEC> v;
            if (use_first) {
                v = first;
            } else {
                v = load_vec<T, VEC>(rp + c);
                RowVec<T, VEC> sv = load_vec<T, VEC>(cp + c);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v.v[i] = from_f32<T>(rndc<DT>(to_f32<T>(v.v[i]) * to_f32<T>(sv.v[i])));
            }
            use_first = false;
            RowVec<T, VEC> o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float qq = quant_code(to_f32<T>(v.v[k]), dv, q.z, DT, DT, qmin, qmax);
                o.v[k] = from_f32<T>(dequant_code(qq, q.s, q.z, DT));
            }
            store_vec<T, VEC>(out + rr * g + c, o);
        }
    }
}

template <typename T>
static int scale_fakequant_t(const void* W, const void* s, int64_t R, int64_t K, int64_t g, int sym, float qmin,
                             float qmax, void* out, hipStream_t st) {
    constexpr int V16 = 16 / sizeof(T);
    const int64_t G = R * (K / g);
    const int gpr = (int)(K / g);
    bool vec_ok = (g % V16 == 0) && (((uintptr_t)W & 15) == 0) && (((uintptr_t)out & 15) == 0) &&
                  (((uintptr_t)s & 15) == 0);
    if (vec_ok && small_ok(g, V16)) {
        int lpr = (int)(g / V16);
        int grid = grid_for(ceil_div64(G, (64 / lpr) * UNR), kBlock / 64);
        hipLaunchKernelGGL((k_quant_dynamic_small<T, V16, LLMC_OUT_FAKE, true>), dim3(grid), dim3(kBlock), 0, st,
                           (const T*)W, (const T*)s, G, (int)g, gpr, lpr, sym, 1, qmin, qmax, out, (T*)nullptr,
                           (T*)nullptr);
    } else if (vec_ok) {
        int lpr = choose_lpr(g, V16);
        int grid = grid_for(ceil_div64(G, 64 / lpr), kBlock / 64);
        hipLaunchKernelGGL((k_scale_fakequant<T, V16>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, (const T*)s, G,
                           (int)g, gpr, lpr, sym, qmin, qmax, (T*)out);
    } else {
        int lpr = choose_lpr(g, 1);
        int grid = grid_for(ceil_div64(G, 64 / lpr), kBlock / 64);
        hipLaunchKernelGGL((k_scale_fakequant<T, 1>), dim3(grid), dim3(kBlock), 0, st, (const T*)W, (const T*)s, G,
                           (int)g, gpr, lpr, sym, qmin, qmax, (T*)out);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
}  // namespace llmc

extern "C" int llmc_awq_scale_fakequant(const void* W, const void* s, int dt, int64_t R, int64_t K, int64_t g,
                                        int sym, float qmin, float qmax, void* out, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt), "awq_scale_fakequant: bad dtype");
    LLMC_REQUIRE(W && s && out && R > 0 && K > 0, "awq_scale_fakequant: null/empty argument");
    if (g <= 0) g = K;
    LLMC_REQUIRE(K % g == 0 && g < (1ll << 31), "awq_scale_fakequant: K must be a multiple of the group size");
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case LLMC_F16: return scale_fakequant_t<f16_t>(W, s, R, K, g, sym, qmin, qmax, out, st);
        case LLMC_BF16: return scale_fakequant_t<bf16_t>(W, s, R, K, g, sym, qmin, qmax, out, st);
        default: return scale_fakequant_t<float>(W, s, R, K, g, sym, qmin, qmax, out, st);
    }
}

extern "C" int llmc_pack_lsb(const void* codes, int code_kind, int64_t R, int64_t K, int bits,
                             int32_t* packed, llmc_stream_t stream) {
    LLMC_REQUIRE(codes && packed && R > 0 && K > 0, "pack_lsb: null/empty argument");
    LLMC_REQUIRE(bits == 4 || bits == 8, "pack_lsb: bits must be 4 or 8");
    LLMC_REQUIRE(code_kind == LLMC_OUT_I32 || code_kind == LLMC_OUT_I8, "pack_lsb: bad code container");
    hipStream_t st = (hipStream_t)stream;
    int pf = 32 / bits;
    int64_t Kp = ceil_div64(K, pf);
    int grid = grid_for(R * Kp, kBlock);
    if (code_kind == LLMC_OUT_I32) {
        bool vec_ok = (K % 4 == 0) && (((uintptr_t)codes & 15) == 0);  // 16-B loads need aligned rows
        if (vec_ok)
            hipLaunchKernelGGL((k_pack_lsb<int32_t, true>), dim3(grid), dim3(kBlock), 0, st,
                               (const int32_t*)codes, R, K, bits, Kp, packed);
        else
            hipLaunchKernelGGL((k_pack_lsb<int32_t, false>), dim3(grid), dim3(kBlock), 0, st,
                               (const int32_t*)codes, R, K, bits, Kp, packed);
    } else {
        hipLaunchKernelGGL((k_pack_lsb<int8_t, false>), dim3(grid), dim3(kBlock), 0, st, (const int8_t*)codes,
                           R, K, bits, Kp, packed);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
