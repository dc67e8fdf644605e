// act_hist.hip — the data pass of `calib_algo: static_hist` (quant.py:462-512): torch.histc of one calibration sample over a
// given range, fp32 arithmetic like `tensor.float()` + histc: bin = int((x - lo) * bins / (hi - lo)), the right edge in
// the last bin, values outside [lo, hi] (and NaN) dropped; lo == hi widens the range by one on both sides like ATen.
// HBM-bound: one read of the sample; counts are taken in LDS (one private histogram per workgroup, integer atomics),
// merged with integer atomics in a workspace and converted to fp32 counts (exact below 2^24 per bin).
#include "common.h"

namespace llmc {

static constexpr int HB = 256;

template <typename T>
__global__ __launch_bounds__(HB) void k_histc(const T* __restrict__ x, int64_t n, int bins, float lo, float hi,
                                              unsigned* __restrict__ cnt) {
    extern __shared__ unsigned hl[];
    for (int i = threadIdx.x; i < bins; i += HB) hl[i] = 0u;
    __syncthreads();
    const float fb = (float)bins, w = hi - lo;
    constexpr int V = 16 / sizeof(T);
    const int64_t nv = n / V;
    for (int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * HB) {
        const uint4 r = *reinterpret_cast<const uint4*>(x + i * V);
        T v[V];
        __builtin_memcpy(v, &r, 16);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float f = to_f32<T>(v[k]);
            if (f >= lo && f <= hi) {
                int p = (int)((f - lo) * fb / w);
                p = p < bins - 1 ? p : bins - 1;
                atomicAdd(&hl[p], 1u);
            }
        }
    }
    if (blockIdx.x == 0)   // ragged tail
        for (int64_t i = nv * V + threadIdx.x; i < n; i += HB) {
            const float f = to_f32<T>(x[i]);
            if (f >= lo && f <= hi) {
                int p = (int)((f - lo) * fb / w);
                p = p < bins - 1 ? p : bins - 1;
                atomicAdd(&hl[p], 1u);
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += HB)
        if (hl[i]) atomicAdd(&cnt[i], hl[i]);
}

__global__ __launch_bounds__(HB) void k_hist_to_f32(const unsigned* __restrict__ cnt, int bins, float* __restrict__ out) {
    const int i = blockIdx.x * HB + threadIdx.x;
    if (i < bins) out[i] = (float)cnt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Per-sample min / max of up to MM_MAX_SAMPLES calibration samples in ONE launch pair (`static_minmax`,
// `static_moving_minmax` and the first pass of `static_hist` take `sample.min()`, `sample.max()` of every sample:
// quant.py:253-263, 524-543, 462-475). The samples are separate allocations (hook outputs); their addresses and lengths travel
// in the kernel arguments like k_syrk4's sample table. HBM-bound: one read of every sample in 16-B pieces.
// NaN: torch's min / max propagate it, fminf / fmaxf drop it, so a NaN anywhere in a sample sets both of its results to NaN.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int MM_MAX_SAMPLES = 160;
static constexpr int MM_CHUNK = 65536;       // elements per workgroup pass

struct MinmaxSamplesArgs {
    int n, nch;              // samples, chunks per sample (of the longest)
    float* part;             // [n][nch][3] min, max, nan flag
    float* mn;               // [n]
    float* mx;               // [n]
    struct { uint64_t base; int64_t len; } smp[MM_MAX_SAMPLES];
};

template <typename T>
__global__ __launch_bounds__(HB) void k_minmax_samples(MinmaxSamplesArgs a) {
    __shared__ float red[3][HB / 64];
    const int row = blockIdx.y, ch = blockIdx.x;
    const T* x = (const T*)(uintptr_t)a.smp[row].base;
    const int64_t n = a.smp[row].len;
    const int64_t c0 = (int64_t)ch * MM_CHUNK, c1 = c0 + MM_CHUNK < n ? c0 + MM_CHUNK : n;
    float mn = INFINITY, mx = -INFINITY, bad = 0.0f;
    constexpr int V = 16 / sizeof(T);
    if (c0 < n) {
        const int64_t v1 = c0 + ((c1 - c0) / V) * V;      // c0 is a multiple of V (MM_CHUNK is), the base is 16-B aligned
        // four 16-byte loads in flight per thread (one per turn left the kernel at 3.8 TB/s on the 18 inputs of a Mixtral block)
        int64_t i = c0 + (int64_t)threadIdx.x * V;
        for (; i + 3 * (int64_t)HB * V < v1; i += 4 * (int64_t)HB * V) {
            uint4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(x + i + u * (int64_t)HB * V);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                T v[V];
                __builtin_memcpy(v, &r[u], 16);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float f = to_f32<T>(v[k]);
                    mn = fminf(mn, f);
                    mx = fmaxf(mx, f);
                    bad = f != f ? 1.0f : bad;
                }
            }
        }
        for (; i < v1; i += (int64_t)HB * V) {
            const uint4 r = *reinterpret_cast<const uint4*>(x + i);
            T v[V];
            __builtin_memcpy(v, &r, 16);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float f = to_f32<T>(v[k]);
                mn = fminf(mn, f);
                mx = fmaxf(mx, f);
                bad = f != f ? 1.0f : bad;
            }
        }
        for (int64_t i = v1 + threadIdx.x; i < c1; i += HB) {      // ragged tail of the last chunk
            const float f = to_f32<T>(x[i]);
            mn = fminf(mn, f);
            mx = fmaxf(mx, f);
            bad = f != f ? 1.0f : bad;
        }
    }
    mn = wave_min(mn, 64);
    mx = wave_max(mx, 64);
    bad = wave_max(bad, 64);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = mn;
        red[1][threadIdx.x >> 6] = mx;
        red[2][threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < HB / 64; ++i) {
            mn = fminf(mn, red[0][i]);
            mx = fmaxf(mx, red[1][i]);
            bad = fmaxf(bad, red[2][i]);
        }
        float* o = a.part + ((int64_t)row * a.nch + ch) * 3;
        o[0] = mn; o[1] = mx; o[2] = bad;
    }
}

__global__ __launch_bounds__(64) void k_minmax_samples_final(MinmaxSamplesArgs a) {
    const int row = blockIdx.x;
    float mn = INFINITY, mx = -INFINITY, bad = 0.0f;
    for (int c = threadIdx.x; c < a.nch; c += 64) {
        const float* o = a.part + ((int64_t)row * a.nch + c) * 3;
        mn = fminf(mn, o[0]);
        mx = fmaxf(mx, o[1]);
        bad = fmaxf(bad, o[2]);
    }
    mn = wave_min(mn, 64);
    mx = wave_max(mx, 64);
    bad = wave_max(bad, 64);
    if (threadIdx.x == 0) {
        const float nanv = __uint_as_float(0x7fc00000u);
        a.mn[row] = bad != 0.0f ? nanv : mn;
        a.mx[row] = bad != 0.0f ? nanv : mx;
    }
}

}  // namespace llmc

using namespace llmc;

#define DISPATCH_DT(dt, CALL)                 \
    switch (dt) {                             \
        case LLMC_F16: { using T = f16_t; CALL; break; }   \
        case LLMC_BF16: { using T = bf16_t; CALL; break; } \
        default: { using T = float; CALL; break; }         \
    }

extern "C" size_t llmc_histc_ws_bytes(int bins) { return bins > 0 ? (size_t)bins * sizeof(unsigned) : 0; }

extern "C" int llmc_histc(const void* x, int dt, int64_t n, int bins, float lo, float hi, float* out, void* ws,
                          llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && x && out && ws && n > 0 && bins > 0 && bins <= 8192, "histc: bad argument");
    LLMC_REQUIRE(((uintptr_t)x & 15) == 0, "histc: x must be 16-B aligned");
    LLMC_REQUIRE(lo <= hi, "histc: min must not exceed max");      // NaN bounds fail here too
    if (lo == hi) { lo -= 1.0f; hi += 1.0f; }
    hipStream_t st = (hipStream_t)stream;
    LLMC_HIP_CHECK(hipMemsetAsync(ws, 0, (size_t)bins * sizeof(unsigned), st));
    const int V = 16 / dtype_size(dt);
    int64_t blocks = ceil_div64(ceil_div64(n, V), HB);
    const int grid = (int)(blocks < 2048 ? (blocks < 1 ? 1 : blocks) : 2048);
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_histc<T>), dim3(grid), dim3(HB), (size_t)bins * sizeof(unsigned), st,
                                       (const T*)x, n, bins, lo, hi, (unsigned*)ws));
    LLMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_hist_to_f32, dim3((bins + HB - 1) / HB), dim3(HB), 0, st, (const unsigned*)ws, bins, out);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_minmax_samples_max(void) { return MM_MAX_SAMPLES; }

extern "C" size_t llmc_minmax_samples_ws_bytes(const int64_t* len_list_host, int n) {
    if (!len_list_host || n <= 0) return 0;
    int64_t longest = 0;
    for (int i = 0; i < n; ++i) longest = len_list_host[i] > longest ? len_list_host[i] : longest;
    return (size_t)n * (size_t)ceil_div64(longest, MM_CHUNK) * 3 * sizeof(float);
}

extern "C" int llmc_minmax_samples(const void* const* X_list_host, const int64_t* len_list_host, int n, int dt, float* mn,
                                   float* mx, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && X_list_host && len_list_host && mn && mx && ws, "minmax_samples: bad argument");
    LLMC_REQUIRE(n >= 1 && n <= MM_MAX_SAMPLES, "minmax_samples: 1 .. llmc_minmax_samples_max() samples per call");
    MinmaxSamplesArgs a;
    int64_t longest = 0;
    for (int i = 0; i < n; ++i) {
        LLMC_REQUIRE(X_list_host[i] && len_list_host[i] > 0 && ((uintptr_t)X_list_host[i] & 15) == 0,
                     "minmax_samples: every sample must be non-empty and 16-B aligned");
        a.smp[i].base = (uint64_t)(uintptr_t)X_list_host[i];
        a.smp[i].len = len_list_host[i];
        longest = len_list_host[i] > longest ? len_list_host[i] : longest;
    }
    a.n = n;
    a.nch = (int)ceil_div64(longest, MM_CHUNK);
    a.part = (float*)ws;
    a.mn = mn;
    a.mx = mx;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(dt, hipLaunchKernelGGL((k_minmax_samples<T>), dim3(a.nch, n), dim3(HB), 0, st, a));
    LLMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax_samples_final, dim3(n), dim3(64), 0, st, a);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
