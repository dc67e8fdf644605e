// fp8_pack.hip — K11 FP8 (OCP e4m3fn) weight quantization and the AutoAWQ GEMM packer (K7b).
//   llmc_fp8_quant      FloatQuantizer sym e4m3 / e5m2: scale = absmax.clamp(1e-5) / finfo.max, q = float_quantize(x / scale)
//                       (quant.py:545-553, 983-1003, 1061-1072, 1195-1221) with qtorch's arithmetic restated (fp8_math.h:
//                       qtorch_quantize; the library is a third-party dependency absent from the reference tree), or
//                       torch's own dtype cast (float8_e4m3fn / float8_e5m2, RNE) for the callers that end in one.
//   llmc_pack_awq_gemm  module_utils.py:1004-1065.
#include "common.h"
#include "fp8_math.h"

namespace llmc {

static constexpr int FB = 256;

// amax[row] = clamp(absmax, 1e-5) in dt (from llmc_minmax_qparams with qmax = 1). scale = amax / 448 in the
// scales' dtype sdt: ATen promotes the 0-dim per-tensor absmax (dt) / 0-dim fp32 qmax to fp32, but keeps dt for
// the per-channel [R,1] absmax.
// mode: bit 0 fake (write dequantized values), bits 4-5 format (0 e4m3, 1 e5m2), bit 8 semantics (0 torch's dtype cast,
// 1 qtorch.float_quantize: fp8_math.h)
static constexpr int FP8_FAKE = 1, FP8_FMT_SHIFT = 4, FP8_QTORCH = 0x100;
template <typename T>
__device__ __forceinline__ void fp8_one(float w, float s, int tdt, int mode, int DT, T* of, uint8_t* ob) {
    const float t = rnd(rnd(w / s, tdt) + 0.0f, tdt);          // tensor / scales + zeros
    float v;
    const uint8_t q = fp8_encode(t, (mode >> FP8_FMT_SHIFT) & 3, mode & FP8_QTORCH, &v);
    if (mode & FP8_FAKE) *of = from_f32<T>(opaque_f32(v * s));   // fp32 product, one rounding to dt
    else *ob = q;
}
// two elements: one hardware conversion (fp8_math.h) on the e4m3 cast path
template <typename T>
__device__ __forceinline__ void fp8_two(float w0, float w1, float s, int tdt, int mode, T* of, uint8_t* ob) {
    const int fake = mode & FP8_FAKE;
    if ((mode & ~FP8_FAKE) == FP8_QTORCH) {      // e4m3 with qtorch's rounding: the quantized values are exact e4m3fn numbers,
        const float v0 = qtorch_quantize<4, 3>(rnd(rnd(w0 / s, tdt) + 0.0f, tdt));     // so the hardware conversion is exact
        const float v1 = qtorch_quantize<4, 3>(rnd(rnd(w1 / s, tdt) + 0.0f, tdt));
        if (fake) {
            of[0] = from_f32<T>(opaque_f32(v0 * s));
            of[1] = from_f32<T>(opaque_f32(v1 * s));
        } else {
            const uint32_t c = f32x2_to_e4m3fn(v0, v1);
            ob[0] = (uint8_t)c;
            ob[1] = (uint8_t)(c >> 8);
        }
        return;
    }
    if (mode & ~FP8_FAKE) {
        fp8_one<T>(w0, s, tdt, mode, 0, &of[0], &ob[0]);
        fp8_one<T>(w1, s, tdt, mode, 0, &of[1], &ob[1]);
        return;
    }
    const float t0 = rnd(rnd(w0 / s, tdt) + 0.0f, tdt), t1 = rnd(rnd(w1 / s, tdt) + 0.0f, tdt);
    const uint32_t c = f32x2_to_e4m3fn(t0, t1);
    const uint8_t q0 = (uint8_t)c, q1 = (uint8_t)(c >> 8);
    if (fake) {
        of[0] = from_f32<T>(opaque_f32(e4m3fn_to_f32(q0) * s));
        of[1] = from_f32<T>(opaque_f32(e4m3fn_to_f32(q1) * s));
    } else {
        ob[0] = q0;
        ob[1] = q1;
    }
}

template <typename T>
__global__ __launch_bounds__(FB) void k_fp8_cast(const T* __restrict__ W, const T* __restrict__ amax, int sdt,
                                                 void* __restrict__ scales, int static_scales, int64_t G, int64_t g,
                                                 int mode, void* __restrict__ out) {
    const int fake = mode & FP8_FAKE;
    const float fmax = fp8_format_max((mode >> FP8_FMT_SHIFT) & 3);
    constexpr int DT = dt_of<T>::value;
    constexpr int V = 16 / sizeof(T);
    const int pdt = promote(DT, sdt);
    const int tdt = (G == 1) ? DT : pdt;   // a 0-dim fp32 scale does not promote the [R,K] tensor, a [R,1] one does
    const bool vec = (g % V == 0) && (((uintptr_t)W & 15) == 0) && (((uintptr_t)out & 15) == 0);
    const int64_t total = G * g;
    if (vec) {
        const int64_t nv = total / V;
        for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * FB) {
            const int64_t e0 = i * V;
            const int64_t row = e0 / g;
            float s;
            if (static_scales) {
                s = load_as_f32(scales, row, sdt);
            } else {
                s = rnd(to_f32<T>(amax[row]) / fmax, sdt);
                if (e0 == row * g) store_from_f32(scales, row, sdt, s);
            }
            if (s == 0.0f) s = 1.0f;                              // scales[scales == 0] = 1 (quant.py:1062)
            uint4 raw = *reinterpret_cast<const uint4*>(W + e0);
            T wv[V];
            __builtin_memcpy(wv, &raw, 16);
            T of[V];
            uint8_t ob[V];
#pragma unroll
            for (int k = 0; k < V; k += 2) fp8_two<T>(to_f32<T>(wv[k]), to_f32<T>(wv[k + 1]), s, tdt, mode, &of[k], &ob[k]);
            if (fake) {
                uint4 o;
                __builtin_memcpy(&o, of, 16);
                *reinterpret_cast<uint4*>((T*)out + e0) = o;
            } else {
                if constexpr (V == 8) {
                    uint2 o;
                    __builtin_memcpy(&o, ob, 8);
                    *reinterpret_cast<uint2*>((uint8_t*)out + e0) = o;
                } else {
                    uint32_t o;
                    __builtin_memcpy(&o, ob, 4);
                    *reinterpret_cast<uint32_t*>((uint8_t*)out + e0) = o;
                }
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < total; i += (int64_t)gridDim.x * FB) {
        const int64_t row = i / g;
        float s;
        if (static_scales) {
            s = load_as_f32(scales, row, sdt);
        } else {
            s = rnd(to_f32<T>(amax[row]) / fmax, sdt);
            if (i == row * g) store_from_f32(scales, row, sdt, s);
        }
        if (s == 0.0f) s = 1.0f;
        T of;
        uint8_t ob;
        fp8_one<T>(to_f32<T>(W[i]), s, tdt, mode, DT, &of, &ob);
        if (fake) ((T*)out)[i] = of; else ((uint8_t*)out)[i] = ob;
    }
}

// AutoAWQ: one thread per output word
template <typename T>
__global__ __launch_bounds__(FB) void k_pack_awq_w(const T* __restrict__ W, const void* __restrict__ scales, int sdt,
                                                   const int32_t* __restrict__ zeros, int64_t R, int64_t K,
                                                   int64_t g, int32_t* __restrict__ qweight) {
    constexpr int WDT = dt_of<T>::value;
    const int p = promote(WDT, LLMC_F16);
    const int64_t R8 = R / 8, ng = K / g;
    const int64_t total = K * R8;
    const int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < total; i += (int64_t)gridDim.x * FB) {
        const int64_t rw = i / K, k = i - rw * K;   // k fastest: coalesced reads of W rows
        const int64_t gi = k / g;
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = rw * 8 + order[j];
            const float s = rnd(load_as_f32(scales, r * ng + gi, sdt), LLMC_F16);   // scales.to(float16)
            const float sz = rnd((float)zeros[r * ng + gi] * s, LLMC_F16);           // zeros * scales
            float t = rnd(to_f32<T>(W[r * K + k]) + sz, p);
            t = rnd(t / s, p);
            const int code = (int)rintf(t);
            word |= ((uint32_t)code) << (4 * j);
        }
        qweight[k * R8 + rw] = (int32_t)word;
    }
}
__global__ __launch_bounds__(FB) void k_pack_awq_z(const void* __restrict__ scales, int sdt,
                                                   const int32_t* __restrict__ zeros, int64_t R, int64_t ng,
                                                   int32_t* __restrict__ qzeros, uint16_t* __restrict__ sout) {
    const int64_t R8 = R / 8;
    const int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < ng * R; i += (int64_t)gridDim.x * FB) {
        const int64_t gi = i / R, r = i - gi * R;
        sout[gi * R + r] = f32_to_f16_bits(load_as_f32(scales, r * ng + gi, sdt));
        if (r < R8) {
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) word |= ((uint32_t)zeros[(r * 8 + order[j]) * ng + gi]) << (4 * j);
            qzeros[gi * R8 + r] = (int32_t)word;
        }
    }
}

static inline int grid_fb(int64_t n) {
    int64_t b = ceil_div64(n, FB);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_fp8_quant_ws_bytes(int64_t G, int64_t g) {
    if (G <= 0 || g <= 0) return 0;
    return (((size_t)G * 4 + 255) & ~(size_t)255) + llmc_minmax_qparams_ws_bytes(G, g);
}

extern "C" int llmc_fp8_quant(const void* W, int dt, int64_t G, int64_t g, int fake, void* out, void* scales,
                              int sdt, int static_scales, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && dtype_ok(sdt) && W && out && scales && G > 0 && g > 0, "fp8_quant: bad argument");
    LLMC_REQUIRE(static_scales || ws, "fp8_quant: workspace required for dynamic scales");
    void* amax = ws;
    if (!static_scales) {
        void* ws2 = (char*)ws + (((size_t)G * 4 + 255) & ~(size_t)255);
        // clamp(absmax, 1e-5) in dt: the symmetric qparams with qmax = 1
        int rc = llmc_minmax_qparams(W, dt, G, g, /*sym*/ 1, 1, -1.0f, 1.0f, amax, nullptr, ws2, stream);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case LLMC_F16:
            hipLaunchKernelGGL((k_fp8_cast<f16_t>), dim3(grid_fb(G * g / 8 + 1)), dim3(FB), 0, st, (const f16_t*)W,
                               (const f16_t*)amax, sdt, scales, static_scales, G, g, fake, out);
            break;
        case LLMC_BF16:
            hipLaunchKernelGGL((k_fp8_cast<bf16_t>), dim3(grid_fb(G * g / 8 + 1)), dim3(FB), 0, st, (const bf16_t*)W,
                               (const bf16_t*)amax, sdt, scales, static_scales, G, g, fake, out);
            break;
        default:
            hipLaunchKernelGGL((k_fp8_cast<float>), dim3(grid_fb(G * g / 4 + 1)), dim3(FB), 0, st, (const float*)W,
                               (const float*)amax, sdt, scales, static_scales, G, g, fake, out);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_pack_awq_gemm(const void* weight, int wdt, const void* scales, int sdt, const int32_t* zeros,
                                  int64_t R, int64_t K, int64_t g, int32_t* qweight, int32_t* qzeros,
                                  void* scales_out_f16, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(wdt) && dtype_ok(sdt), "pack_awq_gemm: bad dtype");
    LLMC_REQUIRE(weight && scales && zeros && qweight && qzeros && scales_out_f16 && R > 0 && K > 0 && g > 0,
                 "pack_awq_gemm: null/empty argument (AutoAWQ needs asymmetric zeros)");
    LLMC_REQUIRE(R % 8 == 0 && K % g == 0, "pack_awq_gemm: R must be a multiple of 8 and K of the group size");
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_fb(K * (R / 8));
    switch (wdt) {
        case LLMC_F16:
            hipLaunchKernelGGL((k_pack_awq_w<f16_t>), dim3(grid), dim3(FB), 0, st, (const f16_t*)weight, scales, sdt,
                               zeros, R, K, g, qweight);
            break;
        case LLMC_BF16:
            hipLaunchKernelGGL((k_pack_awq_w<bf16_t>), dim3(grid), dim3(FB), 0, st, (const bf16_t*)weight, scales, sdt,
                               zeros, R, K, g, qweight);
            break;
        default:
            hipLaunchKernelGGL((k_pack_awq_w<float>), dim3(grid), dim3(FB), 0, st, (const float*)weight, scales, sdt,
                               zeros, R, K, g, qweight);
    }
    LLMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_awq_z, dim3(grid_fb((K / g) * R)), dim3(FB), 0, st, scales, sdt, zeros, R, K / g, qzeros,
                       (uint16_t*)scales_out_f16);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
