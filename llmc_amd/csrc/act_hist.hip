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
