// awq_kernels.hip — HBM-bound pieces of AWQ's scale search (awq.py:48-108) and of apply_scale /
// scaling_input (base_blockwise_quantization.py:597-611, 877-897). All arithmetic in the tensor dtype with
// ATen's per-op rounding; reductions accumulate in fp32 in a fixed order (deterministic).
#include "common.h"

namespace llmc {

static constexpr int AB = 256;
static constexpr int TOK_CHUNK = 512;  // tokens per partial sum of the column mean

// ---- column mean of |x| over tokens: stage 1 partial sums [nchunk][K] fp32, stage 2 ordered sum / N -> dt
template <typename T>
__global__ __launch_bounds__(AB) void k_abs_colsum_partial(const T* __restrict__ X, int64_t N, int64_t K,
                                                           float* __restrict__ part) {
    constexpr int V = 16 / sizeof(T);
    const int64_t c0 = ((int64_t)blockIdx.x * AB + threadIdx.x) * V;
    if (c0 >= K) return;
    const int64_t t0 = (int64_t)blockIdx.y * TOK_CHUNK;
    const int64_t t1 = t0 + TOK_CHUNK < N ? t0 + TOK_CHUNK : N;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.0f;
    const bool vec = c0 + V <= K;
    for (int64_t t = t0; t < t1; ++t) {
        const T* p = X + t * K + c0;
        if (vec) {
            uint4 raw = *reinterpret_cast<const uint4*>(p);
            T v[V];
            __builtin_memcpy(v, &raw, 16);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += fabsf(to_f32<T>(v[i]));
        } else {
            for (int i = 0; i < V && c0 + i < K; ++i) acc[i] += fabsf(to_f32<T>(p[i]));
        }
    }
    for (int i = 0; i < V && c0 + i < K; ++i) part[(int64_t)blockIdx.y * K + c0 + i] = acc[i];
}
template <typename T>
__global__ __launch_bounds__(AB) void k_colsum_final(const float* __restrict__ part, int64_t nchunk, int64_t K,
                                                     float denom, T* __restrict__ out) {
    constexpr int DT = dt_of<T>::value;
    const int64_t c = (int64_t)blockIdx.x * AB + threadIdx.x;
    if (c >= K) return;
    float s = 0.0f;
    for (int64_t i = 0; i < nchunk; ++i) s += part[i * K + c];
    out[c] = from_f32<T>(rndc<DT>(s / denom));
}

// ---- Awq.get_weight_scale for one layer: mean over rows of |w| / groupmax(|w|)   (awq.py:59-66)
// one workgroup per slab of rows; thread owns 16-B column chunks; group max via the LPR-lane shuffle
template <typename T>
__global__ __launch_bounds__(AB) void k_wscale_partial(const T* __restrict__ W, int64_t R, int64_t K, int g,
                                                       int rows_per_block, float* __restrict__ part) {
    constexpr int DT = dt_of<T>::value;
    constexpr int V = 16 / sizeof(T);
    const int lpr = g / V;  // lanes per group (power of two <= 64 enforced by the host)
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    for (int64_t c0 = (int64_t)threadIdx.x * V; c0 < K; c0 += (int64_t)AB * V) {
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.0f;
        for (int64_t r = r0; r < r1; ++r) {
            uint4 raw = *reinterpret_cast<const uint4*>(W + r * K + c0);
            T v[V];
            __builtin_memcpy(v, &raw, 16);
            float a[V], m = 0.0f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                a[i] = fabsf(to_f32<T>(v[i]));
                m = fmaxf(m, a[i]);
            }
            m = wave_max(m, lpr);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += rndc<DT>(a[i] / m);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) part[(int64_t)blockIdx.x * K + c0 + i] = acc[i];
    }
}

// ---- Awq.get_scales (awq.py:98-108), single workgroup
template <typename T>
__global__ __launch_bounds__(1024) void k_awq_scales(const T* __restrict__ xm, const T* __restrict__ wm, int64_t K,
                                                     float ratio, float one_minus_ratio, int version,
                                                     T* __restrict__ out) {
    constexpr int DT = dt_of<T>::value;
    __shared__ float smx[16], smn[16];
    const float lo = rndc<DT>(1e-4f);
    // Tensor.pow(python_float) casts the exponent to the tensor dtype first (ATen CPU scalar handling)
    const float er = rndc<DT>(ratio), ew = rndc<DT>(one_minus_ratio);
    float mx = -INFINITY, mn = INFINITY;
    for (int64_t k = threadIdx.x; k < K; k += 1024) {
        float s;
        const float x = to_f32<T>(xm[k]);
        if (version == 1) {
            const float a = rndc<DT>(powf(x, er));
            const float b = rndc<DT>(powf(to_f32<T>(wm[k]), ew));
            s = rndc<DT>(a / b);
        } else {
            s = rndc<DT>(powf(x, er));
        }
        s = fmaxf(s, lo);
        out[k] = from_f32<T>(s);
        mx = fmaxf(mx, s);
        mn = fminf(mn, s);
    }
    mx = wave_max(mx, 64);
    mn = wave_min(mn, 64);
    if ((threadIdx.x & 63) == 0) {
        smx[threadIdx.x >> 6] = mx;
        smn[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    mx = smx[0];
    mn = smn[0];
    for (int i = 1; i < 16; ++i) {
        mx = fmaxf(mx, smx[i]);
        mn = fminf(mn, smn[i]);
    }
    const float den = rndc<DT>(sqrtf(rndc<DT>(mx * mn)));
    for (int64_t k = threadIdx.x; k < K; k += 1024) out[k] = from_f32<T>(rndc<DT>(to_f32<T>(out[k]) / den));
}

// ---- x / s[col], w * s[col], clamp per (row, group)
template <typename T, int OP>  // OP 0: out = x / s ; 1: out = x * s
__global__ __launch_bounds__(AB) void k_cols_op(const T* __restrict__ X, const T* __restrict__ s, int64_t N,
                                                int64_t K, T* __restrict__ out) {
    constexpr int DT = dt_of<T>::value;
    constexpr int V = 16 / sizeof(T);
    const int64_t nv = K / V;
    const int64_t total = N * nv;
    for (int64_t i = (int64_t)blockIdx.x * AB + threadIdx.x; i < total; i += (int64_t)gridDim.x * AB) {
        const int64_t r = i / nv, c = (i - r * nv) * V;
        uint4 rx = *reinterpret_cast<const uint4*>(X + r * K + c);
        uint4 rs = *reinterpret_cast<const uint4*>(s + c);
        T xv[V], sv[V], ov[V];
        __builtin_memcpy(xv, &rx, 16);
        __builtin_memcpy(sv, &rs, 16);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float a = to_f32<T>(xv[k]), b = to_f32<T>(sv[k]);
            ov[k] = from_f32<T>(rndc<DT>(OP == 0 ? a / b : a * b));
        }
        uint4 ro;
        __builtin_memcpy(&ro, ov, 16);
        *reinterpret_cast<uint4*>(out + r * K + c) = ro;
    }
}
// x / s[col] written K-TILED (T[K/32][rows][32], linear_eval.hip): block = 64 rows x two 32-k slices, thread = one 16-B
// chunk of each; whole 128-B lines in, KiB runs out. 16-bit T only.
template <typename T>
__global__ __launch_bounds__(256) void k_div_cols_kt(const T* __restrict__ X, const T* __restrict__ s, int64_t N,
                                                     int64_t K, T* __restrict__ out) {
    constexpr int DT = dt_of<T>::value;
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    const int64_t row = (int64_t)blockIdx.x * 64 + r;
    if (row >= N) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t kt = (int64_t)blockIdx.y * 2 + j;
        if (kt * 32 >= K) break;
        const int64_t col = kt * 32 + c * 8;
        uint4 rx = *reinterpret_cast<const uint4*>(X + row * K + col);
        uint4 rs = *reinterpret_cast<const uint4*>(s + col);
        T xv[8], sv[8], ov[8];
        __builtin_memcpy(xv, &rx, 16);
        __builtin_memcpy(sv, &rs, 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) ov[k] = from_f32<T>(rndc<DT>(to_f32<T>(xv[k]) / to_f32<T>(sv[k])));
        uint4 ro;
        __builtin_memcpy(&ro, ov, 16);
        *reinterpret_cast<uint4*>(out + (kt * N + row) * 32 + c * 8) = ro;
    }
}
template <typename T>
__global__ __launch_bounds__(AB) void k_clamp_groups(T* __restrict__ W, int64_t R, int64_t K, int64_t g,
                                                     const T* __restrict__ mn, const T* __restrict__ mx) {
    const int64_t total = R * K;
    const int64_t gpr = K / g;
    for (int64_t i = (int64_t)blockIdx.x * AB + threadIdx.x; i < total; i += (int64_t)gridDim.x * AB) {
        const int64_t r = i / K, c = i - r * K;
        const int64_t gi = r * gpr + c / g;
        float v = to_f32<T>(W[i]);
        v = fminf(fmaxf(v, to_f32<T>(mn[gi])), to_f32<T>(mx[gi]));
        W[i] = from_f32<T>(v);
    }
}

static inline int grid1d(int64_t items) {
    int64_t b = ceil_div64(items, AB);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace llmc

using namespace llmc;

#define DISPATCH_DT(dt, CALL)                 \
    switch (dt) {                             \
        case LLMC_F16: { using T = f16_t; CALL; break; }   \
        case LLMC_BF16: { using T = bf16_t; CALL; break; } \
        default: { using T = float; CALL; break; }         \
    }

extern "C" size_t llmc_awq_act_mean_ws_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    return (size_t)ceil_div64(N, TOK_CHUNK) * K * sizeof(float);
}

extern "C" int llmc_awq_act_mean(const void* X, int dt, int64_t N, int64_t K, void* out, void* ws,
                                 llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt), "awq_act_mean: bad dtype");
    LLMC_REQUIRE(X && out && ws && N > 0 && K > 0, "awq_act_mean: null/empty argument");
    LLMC_REQUIRE(((uintptr_t)X & 15) == 0 && (K * dtype_size(dt)) % 16 == 0, "awq_act_mean: rows must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nchunk = ceil_div64(N, TOK_CHUNK);
    const int V = 16 / dtype_size(dt);
    dim3 grid((unsigned)ceil_div64(K, (int64_t)AB * V), (unsigned)nchunk);
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_abs_colsum_partial<T>), grid, dim3(AB), 0, st, (const T*)X, N, K, (float*)ws));
    LLMC_LAUNCH_CHECK();
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_colsum_final<T>), dim3((unsigned)ceil_div64(K, AB)), dim3(AB), 0, st,
                                       (const float*)ws, nchunk, K, (float)N, (T*)out));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" size_t llmc_awq_weight_mean_ws_bytes(int64_t R, int64_t K) {
    if (R <= 0 || K <= 0) return 0;
    return (size_t)ceil_div64(R, 16) * K * sizeof(float);
}

extern "C" int llmc_awq_weight_mean(const void* W, int dt, int64_t R, int64_t K, int64_t g, void* out_dt, void* ws,
                                    llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt), "awq_weight_mean: bad dtype");
    LLMC_REQUIRE(W && out_dt && ws && R > 0 && K > 0, "awq_weight_mean: null/empty argument");
    if (g <= 0) g = K;
    const int V = 16 / dtype_size(dt);
    const int64_t lpr = g / V;
    if (!(K % g == 0 && g % V == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && K % ((int64_t)V) == 0 &&
          ((uintptr_t)W & 15) == 0)) {
        set_last_error_msg("awq_weight_mean: group size must be 16 B x (power of two <= 64) elements");
        return LLMC_ENOTSUP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 16;
    const int64_t nblk = ceil_div64(R, rpb);
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_wscale_partial<T>), dim3((unsigned)nblk), dim3(AB), 0, st, (const T*)W, R, K,
                                       (int)g, rpb, (float*)ws));
    LLMC_LAUNCH_CHECK();
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_colsum_final<T>), dim3((unsigned)ceil_div64(K, AB)), dim3(AB), 0, st,
                                       (const float*)ws, nblk, K, (float)R, (T*)out_dt));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_awq_scales(const void* x_mean, const void* w_mean, int dt, int64_t K, double ratio_d,
                               int version, void* out, llmc_stream_t stream) {
    const float ratio = (float)ratio_d;
    LLMC_REQUIRE(dtype_ok(dt), "awq_scales: bad dtype");
    LLMC_REQUIRE(x_mean && out && K > 0, "awq_scales: null/empty argument");
    LLMC_REQUIRE(version == 2 || (version == 1 && w_mean), "awq_scales: v1 needs w_mean");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_awq_scales<T>), dim3(1), dim3(1024), 0, st, (const T*)x_mean,
                                       (const T*)w_mean, K, ratio, (float)(1.0 - (double)ratio_d), version, (T*)out));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_div_cols(const void* X, const void* s, int dt, int64_t N, int64_t K, void* out,
                             llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && X && s && out && N > 0 && K > 0, "div_cols: bad argument");
    LLMC_REQUIRE((K * dtype_size(dt)) % 16 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                     ((uintptr_t)s & 15) == 0, "div_cols: rows must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int V = 16 / dtype_size(dt);
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_cols_op<T, 0>), dim3(grid1d(N * (K / V))), dim3(AB), 0, st, (const T*)X,
                                       (const T*)s, N, K, (T*)out));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_div_cols_kt(const void* X, const void* s, int dt, int64_t N, int64_t K, void* out,
                                llmc_stream_t stream) {
    LLMC_REQUIRE((dt == LLMC_F16 || dt == LLMC_BF16) && X && s && out && X != out && N > 0 && K > 0,
                 "div_cols_kt: bad argument (16-bit dtypes, out of place)");
    LLMC_REQUIRE(K % 32 == 0 && K / 64 < 65535 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                     ((uintptr_t)s & 15) == 0, "div_cols_kt: needs K % 32 == 0 and 16-B aligned buffers");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)ceil_div64(K, 64));
    if (dt == LLMC_F16) hipLaunchKernelGGL((k_div_cols_kt<f16_t>), grid, dim3(256), 0, st, (const f16_t*)X, (const f16_t*)s, N, K, (f16_t*)out);
    else hipLaunchKernelGGL((k_div_cols_kt<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)s, N, K, (bf16_t*)out);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_mul_cols(void* W, const void* s, int dt, int64_t R, int64_t K, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && W && s && R > 0 && K > 0, "mul_cols: bad argument");
    LLMC_REQUIRE((K * dtype_size(dt)) % 16 == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)s & 15) == 0,
                 "mul_cols: rows must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int V = 16 / dtype_size(dt);
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_cols_op<T, 1>), dim3(grid1d(R * (K / V))), dim3(AB), 0, st, (const T*)W,
                                       (const T*)s, R, K, (T*)W));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_clamp_groups(void* W, int dt, int64_t R, int64_t K, int64_t g, const void* min_val,
                                 const void* max_val, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && W && min_val && max_val && R > 0 && K > 0, "clamp_groups: bad argument");
    if (g <= 0) g = K;
    LLMC_REQUIRE(K % g == 0, "clamp_groups: K must be a multiple of the group size");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_clamp_groups<T>), dim3(grid1d(R * K)), dim3(AB), 0, st, (T*)W, R, K, g,
                                       (const T*)min_val, (const T*)max_val));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
