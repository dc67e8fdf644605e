// fp8_block.hip — FP8 (OCP e4m3fn) block-wise path: the DeepSeek-V3 style 128x128 weight blocks / 1x128 activation
// blocks of llmc's Triton kernels (llmc/compression/quantization/kernel.py:7-242) and of FloatQuantizer's
// `per_block` granularity (quant.py:18-43, 132-143, 612-658, 1043-1072, 1161-1221), as HIP kernels for gfx950:
//   llmc_fp8_block_quant    weight_cast_to_fp8 / FloatQuantizer per_block fake + real quant
//   llmc_fp8_block_dequant  weight_cast_to_bf16
//   llmc_fp8_act_quant      act_quant (one scale per `block` consecutive elements of a row)
//   llmc_fp8_block_gemm     fp8_gemm + block_wise_fp8_forward_func: C = sum_kb (A_kb . B_kb^T) * a_s[m, kb] * b_s[n/128, kb]
//                           on v_mfma_f32_32x32x16_fp8_fp8 (gfx950 reads OCP e4m3fn, not MI300's fnuz)
// e4m3 rounding is pinned to torch.float8_e4m3fn's cast (RNE) like fp8_pack.hip; products of two e4m3 values are
// exact in fp32, the 128-deep partial sums accumulate in fp32 inside the MFMA.
#include "common.h"
#include "mfma_common.h"

namespace llmc {

// fp32 <-> e4m3fn with torch's semantics (same arithmetic as fp8_pack.hip)
__device__ __forceinline__ uint8_t fb_f32_to_e4m3fn(float x) {
    const uint32_t b = __float_as_uint(x);
    const uint32_t sign = (b >> 24) & 0x80u;
    const uint32_t ab = b & 0x7fffffffu;
    if (ab > 0x7f800000u) return (uint8_t)(sign | 0x7f);
    const float ax = __uint_as_float(ab);
    if (ax < 0.015625f) return (uint8_t)(sign | (uint32_t)rintf(ax * 512.0f));
    uint32_t r = ab + 0x7ffffu + ((ab >> 20) & 1u);
    r &= 0xfff00000u;
    if (r > 0x43e00000u) return (uint8_t)(sign | 0x7f);
    return (uint8_t)(sign | (((r >> 23) - 120u) << 3) | ((r >> 20) & 7u));
}
__device__ __forceinline__ float fb_e4m3fn_to_f32(uint8_t v) {
    const uint32_t e = (v >> 3) & 0xf, m = v & 7;
    float r;
    if ((v & 0x7f) == 0x7f) r = __uint_as_float(0x7fc00000u);
    else if (e == 0) r = (float)m * 0.001953125f;
    else r = __uint_as_float(((e + 120u) << 23) | (m << 20));
    // sign by bit: a negative zero must stay negative ((q - 0) * s = -0.0 in the reference)
    return __uint_as_float(__float_as_uint(r) | ((uint32_t)(v & 0x80) << 24));
}

// ---------------------------------------------------------------------------------------------------------------
// One workgroup per b x b block (b <= 128): absmax, scale = max(absmax, clamp_min) / 448 in fp32, then
//   q = e4m3(x / scale)   (fp32 division: a [M/b,1,N/b,1] fp32 scale promotes the 16-bit tensor, quant.py:1063)
// real: q bytes; fake: q * scale rounded once to the tensor dtype.
// clamp_min = 1e-5 is FloatQuantizer's `.clamp(min=1e-5)` (quant.py:551); 0 is kernel.py's weight_cast_to_fp8,
// where an all-zero block gives scale 0 and x / 0 = NaN exactly like the Triton kernel.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_block_quant(const T* __restrict__ W, int64_t M, int64_t N, int b,
                                                         float clamp_min, int zero_scale_to_one, int fake,
                                                         void* __restrict__ out, float* __restrict__ scales) {
    __shared__ float red[4];
    const int64_t r0 = (int64_t)blockIdx.y * b, c0 = (int64_t)blockIdx.x * b;
    const int nbn = gridDim.x;
    const int tid = threadIdx.x;
    float am = 0.0f;
    for (int i = tid; i < b * b; i += 256) {
        const int64_t r = r0 + i / b, c = c0 + i % b;
        if (r < M && c < N) am = fmaxf(am, fabsf(to_f32<T>(W[r * N + c])));
    }
    am = wave_max(am, 64);
    if ((tid & 63) == 0) red[tid >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s;
    if (fake & 2) s = scales[(int64_t)blockIdx.y * nbn + blockIdx.x];      // static: the caller's scales
    else {
        s = fmaxf(am, clamp_min) / 448.0f;
        if (tid == 0) scales[(int64_t)blockIdx.y * nbn + blockIdx.x] = s;
    }
    if (zero_scale_to_one && s == 0.0f) s = 1.0f;          // scales[scales == 0] = 1 (quant.py:1062)
    for (int i = tid; i < b * b; i += 256) {
        const int64_t r = r0 + i / b, c = c0 + i % b;
        if (r < M && c < N) {
            const float x = to_f32<T>(W[r * N + c]);
            const uint8_t q = fb_f32_to_e4m3fn(x / s + 0.0f);
            if (fake & 1) ((T*)out)[r * N + c] = from_f32<T>(opaque_f32(fb_e4m3fn_to_f32(q) * s));
            else ((uint8_t*)out)[r * N + c] = q;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_fp8_block_dequant(const uint8_t* __restrict__ W, const float* __restrict__ scales,
                                                           int64_t M, int64_t N, int b, int nbn, T* __restrict__ out) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N, c = i - r * N;
        out[i] = from_f32<T>(opaque_f32(fb_e4m3fn_to_f32(W[i]) * scales[(r / b) * nbn + c / b]));
    }
}

// act_quant: 16 lanes per block of b <= 128 elements (8 per lane), four blocks per wave
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_act_quant(const T* __restrict__ X, int64_t nblocks, int b,
                                                       uint8_t* __restrict__ Y, float* __restrict__ S) {
    const int sub = threadIdx.x & 15;
    const int64_t blk = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (blk >= nblocks) return;
    const T* x = X + blk * b;
    float v[8];
    float am = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = sub * 8 + e;
        v[e] = c < b ? to_f32<T>(x[c]) : 0.0f;
        am = fmaxf(am, fabsf(v[e]));
    }
    am = wave_max(am, 16);
    const float s = am / 448.0f;                              // kernel.py:24 (no clamp: a zero block divides 0 / 0)
    if (sub == 0) S[blk] = s;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = sub * 8 + e;
        if (c < b) Y[blk * b + c] = fb_f32_to_e4m3fn(v[e] / s);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fp8 GEMM with 128-deep block scales: A [M, K] e4m3 row-major + a_s [M, K/128], B [N, K] e4m3 row-major (a weight)
// + b_s [N/128, K/128]; C [M, N] in dt. Workgroup tile 128 x 128 (one b_s per K block), 4 waves as 2 x 2, each
// 64 x 64 = 2 x 2 MFMA 32x32x16 accumulators. Per K block of 128: both 128 x 128-byte panels are staged in LDS
// (rows padded to 144 B: the 8-byte fragment reads of a 32-lane half then touch every bank pair once), eight
// fp8 MFMAs per accumulator build the block's partial sum, which enters the result as
//     acc += (partial * a_s[row]) * b_s          (the order of kernel.py:226; no contraction)
// ---------------------------------------------------------------------------------------------------------------
static constexpr int FG_T = 128;
static constexpr int FG_LD = 144;

template <int DT>
__global__ __launch_bounds__(256) void k_fp8_block_gemm(const uint8_t* __restrict__ A, const float* __restrict__ As,
                                                        const uint8_t* __restrict__ B, const float* __restrict__ Bs,
                                                        int64_t M, int64_t N, int64_t K, const void* __restrict__ bias,
                                                        void* __restrict__ C) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * FG_T * FG_LD];
    uint8_t* la = lds;
    uint8_t* lb = lds + FG_T * FG_LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int64_t m0 = (int64_t)blockIdx.y * FG_T, n0 = (int64_t)blockIdx.x * FG_T;
    const int nkb = (int)((K + 127) / 128);
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // staging: thread t loads 16 B chunk (t & 7) of rows (t >> 3) + 32 q, q = 0..3
    const int srow = tid >> 3, sch = (tid & 7) * 16;
    for (int kb = 0; kb < nkb; ++kb) {
        const int64_t k0 = (int64_t)kb * 128;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = srow + 32 * q;
            uint4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
            const int64_t kk = k0 + sch;
            if (m0 + row < M) {
                const uint8_t* p = A + (m0 + row) * K + kk;
                if (kk + 16 <= K && (((uintptr_t)p) & 15) == 0) va = *reinterpret_cast<const uint4*>(p);
                else {
                    uint8_t t[16];
                    for (int e = 0; e < 16; ++e) t[e] = kk + e < K ? p[e] : 0;
                    __builtin_memcpy(&va, t, 16);
                }
            }
            if (n0 + row < N) {
                const uint8_t* p = B + (n0 + row) * K + kk;
                if (kk + 16 <= K && (((uintptr_t)p) & 15) == 0) vb = *reinterpret_cast<const uint4*>(p);
                else {
                    uint8_t t[16];
                    for (int e = 0; e < 16; ++e) t[e] = kk + e < K ? p[e] : 0;
                    __builtin_memcpy(&vb, t, 16);
                }
            }
            *reinterpret_cast<uint4*>(la + row * FG_LD + sch) = va;
            *reinterpret_cast<uint4*>(lb + row * FG_LD + sch) = vb;
        }
        __syncthreads();
        f32x16 part[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[m][n][r] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            long fa[2], fb[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                fa[m] = *reinterpret_cast<const long*>(la + (wm * 64 + m * 32 + (lane & 31)) * FG_LD + kc * 16 + 8 * (lane >> 5));
#pragma unroll
            for (int n = 0; n < 2; ++n)
                fb[n] = *reinterpret_cast<const long*>(lb + (wn * 64 + n * 32 + (lane & 31)) * FG_LD + kc * 16 + 8 * (lane >> 5));
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    part[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(fa[m], fb[n], part[m][n], 0, 0, 0);
        }
        // block scales: a_s per output row, b_s per (128-row block of B, K block)
        const float bs = Bs[(n0 / 128) * nkb + kb];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float as = row < M ? As[row * nkb + kb] : 0.0f;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    float t = part[m][n][r] * as;
                    t = t * bs;
                    acc[m][n][r] = acc[m][n][r] + t;
                }
            }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int64_t col = n0 + wn * 64 + n * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M && col < N) {
                    float y = rndc<DT>(acc[m][n][r]);                       // accumulator.to(c dtype)
                    if (bias) y = rndc<DT>(y + load_as_f32(bias, col, DT));   // y += bias in the output dtype
                    store_from_f32(C, row * N + col, DT, y);
                }
            }
        }
}

}  // namespace llmc

using namespace llmc;

extern "C" int llmc_fp8_block_quant(const void* W, int dt, int64_t M, int64_t N, int block, float clamp_min,
                                    int fake, void* out, float* scales, llmc_stream_t stream) {
    LLMC_REQUIRE(W && out && scales && M > 0 && N > 0, "fp8_block_quant: null/empty argument");
    LLMC_REQUIRE(dtype_ok(dt), "fp8_block_quant: bad dtype");
    LLMC_REQUIRE(block >= 16 && block <= 128, "fp8_block_quant: block size must be in [16, 128]");
    dim3 grid((unsigned)ceil_div64(N, block), (unsigned)ceil_div64(M, block));
    hipStream_t st = (hipStream_t)stream;
    const int z1 = clamp_min > 0.0f ? 1 : 0;   // FloatQuantizer semantics replace a zero scale, kernel.py's do not
    switch (dt) {
        case LLMC_F16: hipLaunchKernelGGL((k_fp8_block_quant<f16_t>), grid, dim3(256), 0, st, (const f16_t*)W, M, N, block, clamp_min, z1, fake, out, scales); break;
        case LLMC_BF16: hipLaunchKernelGGL((k_fp8_block_quant<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)W, M, N, block, clamp_min, z1, fake, out, scales); break;
        default: hipLaunchKernelGGL((k_fp8_block_quant<float>), grid, dim3(256), 0, st, (const float*)W, M, N, block, clamp_min, z1, fake, out, scales); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_fp8_block_dequant(const void* W8, const float* scales, int64_t M, int64_t N, int block, int out_dt,
                                      void* out, llmc_stream_t stream) {
    LLMC_REQUIRE(W8 && out && scales && M > 0 && N > 0, "fp8_block_dequant: null/empty argument");
    LLMC_REQUIRE(dtype_ok(out_dt) && block >= 1, "fp8_block_dequant: bad dtype / block");
    const int nbn = (int)ceil_div64(N, block);
    const int64_t total = M * N;
    const int grid = (int)(ceil_div64(total, 256) < 65536 ? ceil_div64(total, 256) : 65536);
    hipStream_t st = (hipStream_t)stream;
    switch (out_dt) {
        case LLMC_F16: hipLaunchKernelGGL((k_fp8_block_dequant<f16_t>), dim3(grid), dim3(256), 0, st, (const uint8_t*)W8, scales, M, N, block, nbn, (f16_t*)out); break;
        case LLMC_BF16: hipLaunchKernelGGL((k_fp8_block_dequant<bf16_t>), dim3(grid), dim3(256), 0, st, (const uint8_t*)W8, scales, M, N, block, nbn, (bf16_t*)out); break;
        default: hipLaunchKernelGGL((k_fp8_block_dequant<float>), dim3(grid), dim3(256), 0, st, (const uint8_t*)W8, scales, M, N, block, nbn, (float*)out); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_fp8_act_quant(const void* X, int dt, int64_t n_elem, int block, void* out8, float* scales,
                                  llmc_stream_t stream) {
    LLMC_REQUIRE(X && out8 && scales && n_elem > 0, "fp8_act_quant: null/empty argument");
    LLMC_REQUIRE(dtype_ok(dt), "fp8_act_quant: bad dtype");
    LLMC_REQUIRE(block >= 8 && block <= 128 && n_elem % block == 0, "fp8_act_quant: last dim must be a multiple of block <= 128");
    const int64_t nblocks = n_elem / block;
    const int grid = (int)ceil_div64(nblocks * 16, 256);
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case LLMC_F16: hipLaunchKernelGGL((k_fp8_act_quant<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)X, nblocks, block, (uint8_t*)out8, scales); break;
        case LLMC_BF16: hipLaunchKernelGGL((k_fp8_act_quant<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)X, nblocks, block, (uint8_t*)out8, scales); break;
        default: hipLaunchKernelGGL((k_fp8_act_quant<float>), dim3(grid), dim3(256), 0, st, (const float*)X, nblocks, block, (uint8_t*)out8, scales); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_fp8_block_gemm(const void* A8, const float* a_s, const void* B8, const float* b_s, int64_t M,
                                   int64_t N, int64_t K, int out_dt, const void* bias, void* C, llmc_stream_t stream) {
    LLMC_REQUIRE(A8 && a_s && B8 && b_s && C && M > 0 && N > 0 && K > 0, "fp8_block_gemm: null/empty argument");
    LLMC_REQUIRE(out_dt == LLMC_F16 || out_dt == LLMC_BF16 || out_dt == LLMC_F32, "fp8_block_gemm: bad output dtype");
    dim3 grid((unsigned)ceil_div64(N, FG_T), (unsigned)ceil_div64(M, FG_T));
    hipStream_t st = (hipStream_t)stream;
    switch (out_dt) {
        case LLMC_F16: hipLaunchKernelGGL((k_fp8_block_gemm<LLMC_F16>), grid, dim3(256), 0, st, (const uint8_t*)A8, a_s, (const uint8_t*)B8, b_s, M, N, K, bias, C); break;
        case LLMC_BF16: hipLaunchKernelGGL((k_fp8_block_gemm<LLMC_BF16>), grid, dim3(256), 0, st, (const uint8_t*)A8, a_s, (const uint8_t*)B8, b_s, M, N, K, bias, C); break;
        default: hipLaunchKernelGGL((k_fp8_block_gemm<LLMC_F32>), grid, dim3(256), 0, st, (const uint8_t*)A8, a_s, (const uint8_t*)B8, b_s, M, N, K, bias, C); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
