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
#include <stdlib.h>
#include <type_traits>
#include "fp8_math.h"
#include "quant_math.h"
#include "mfma_common.h"

namespace llmc {

// fp32 <-> e4m3fn: fp8_math.h (torch's semantics; the pair form uses the hardware conversion for in-range values)
__device__ __forceinline__ uint8_t fb_f32_to_e4m3fn(float x) { return f32_to_e4m3fn(x); }
__device__ __forceinline__ uint32_t fb_f32x2_to_e4m3fn(float x, float y) { return f32x2_to_e4m3fn(x, y); }
__device__ __forceinline__ float fb_e4m3fn_to_f32(uint8_t v) { return e4m3fn_to_f32(v); }

// ---------------------------------------------------------------------------------------------------------------
// One workgroup per b x b block (b <= 128): absmax, scale = max(absmax, clamp_min) / 448 in fp32, then
//   q = e4m3(x / scale)   (fp32 division: a [M/b,1,N/b,1] fp32 scale promotes the 16-bit tensor, quant.py:1063)
// real: q bytes; fake: q * scale rounded once to the tensor dtype.
// clamp_min = 1e-5 is FloatQuantizer's `.clamp(min=1e-5)` (quant.py:551); 0 is kernel.py's weight_cast_to_fp8,
// where an all-zero block gives scale 0 and x / 0 = NaN exactly like the Triton kernel.
// ---------------------------------------------------------------------------------------------------------------
// eight consecutive elements of a 16-byte aligned run (16-bit types: one 16-B load; fp32: two)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = to_f32<T>(T{(uint16_t)(w[e] & 0xffffu)});
            v[2 * e + 1] = to_f32<T>(T{(uint16_t)(w[e] >> 16)});
        }
    } else {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)from_f32<T>(v[2 * e]).u | ((uint32_t)from_f32<T>(v[2 * e + 1]).u << 16);
        *reinterpret_cast<uint4*>(p) = uint4{w[0], w[1], w[2], w[3]};
    } else {
        reinterpret_cast<float4*>(p)[0] = float4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<float4*>(p)[1] = float4{v[4], v[5], v[6], v[7]};
    }
}
__device__ __forceinline__ void store8_bytes(uint8_t* p, const uint8_t (&q)[8]) {
    uint2 o;
    o.x = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    o.y = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
    *reinterpret_cast<uint2*>(p) = o;
}

// The 128 x 128 interior blocks of an aligned tensor (N % 8 == 0, 16-B aligned base): a thread keeps its 64 elements
// (8 rows x 8 consecutive columns) in registers between the absmax pass and the cast, so the block is read once, in 16-B
// pieces, and written in 8-B (codes) / 16-B (fake) pieces. Edge blocks and other block sizes run k_fp8_block_quant.
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_block_quant128(const T* __restrict__ W, int64_t M, int64_t N, float clamp_min,
                                                            int zero_scale_to_one, int fake, void* __restrict__ out,
                                                            float* __restrict__ scales, int nbn) {
    __shared__ float red[4];
    const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 128;
    const int tid = threadIdx.x;
    const int64_t col = c0 + (tid & 15) * 8;
    float v[8][8];
    float am = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        load8<T>(W + (r0 + (tid >> 4) + 16 * j) * N + col, v[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[j][e]));
    }
    am = wave_max(am, 64);
    if ((tid & 63) == 0) red[tid >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s;
    if (fake & 2) s = scales[(int64_t)blockIdx.y * nbn + blockIdx.x];
    else {
        s = fmaxf(am, clamp_min) / 448.0f;
        if (tid == 0) scales[(int64_t)blockIdx.y * nbn + blockIdx.x] = s;
    }
    if (zero_scale_to_one && s == 0.0f) s = 1.0f;
    // x / s for the block's 16384 elements: with a plain divisor and plain numerators the IEEE quotient is the five-op tail
    // of the division sequence on the refined reciprocal, bit for bit (quant_math.h: rcp_refined, div_tail)
    const bool plain = plain_pos(s) && am < 0x1p40f && !(fake & 0x200);
    const float yr = rcp_refined(s);
    auto quot = [&](float x) { return (plain ? div_tail(x, s, yr) : x / s) + 0.0f; };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t off = (r0 + (tid >> 4) + 16 * j) * N + col;
        uint8_t q[8];
        float dq[8];
        if (fake & 0x100) {          // FloatQuantizer: qtorch.float_quantize (fp8_math.h), not the dtype cast
#pragma unroll
            for (int e = 0; e < 8; ++e) dq[e] = qtorch_e4m3_select(quot(v[j][e]));
#pragma unroll
            for (int e = 0; e < 8; e += 2) {                 // the quantized values are e4m3 numbers: the conversion is exact
                const uint32_t c = fb_f32x2_to_e4m3fn(dq[e], dq[e + 1]);
                q[e] = (uint8_t)c;
                q[e + 1] = (uint8_t)(c >> 8);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t c = fb_f32x2_to_e4m3fn(quot(v[j][e]), quot(v[j][e + 1]));
                q[e] = (uint8_t)c;
                q[e + 1] = (uint8_t)(c >> 8);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) dq[e] = fb_e4m3fn_to_f32(q[e]);
        }
        if (fake & 1) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = opaque_f32(dq[e] * s);
            store8<T>((T*)out + off, y);
        } else store8_bytes((uint8_t*)out + off, q);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_fp8_block_quant(const T* __restrict__ W, int64_t M, int64_t N, int b,
                                                         float clamp_min, int zero_scale_to_one, int fake,
                                                         void* __restrict__ out, float* __restrict__ scales,
                                                         int nbn, int edge) {
    // edge = 0: every block, grid (nbn, nbm). edge = 1: the last block column (grid (1, nbm), ragged N); edge = 2: the last
    // block row (grid (nbn, 1), ragged M) — what k_fp8_block_quant128 leaves over
    __shared__ float red[4];
    const int bx = edge == 1 ? nbn - 1 : blockIdx.x;
    const int by = edge == 2 ? (int)((M + b - 1) / b) - 1 : blockIdx.y;
    const int64_t r0 = (int64_t)by * b, c0 = (int64_t)bx * b;
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
    if (fake & 2) s = scales[(int64_t)by * nbn + bx];      // static: the caller's scales
    else {
        s = fmaxf(am, clamp_min) / 448.0f;
        if (tid == 0) scales[(int64_t)by * nbn + bx] = s;
    }
    if (zero_scale_to_one && s == 0.0f) s = 1.0f;          // scales[scales == 0] = 1 (quant.py:1062)
    for (int i = tid; i < b * b; i += 256) {
        const int64_t r = r0 + i / b, c = c0 + i % b;
        if (r < M && c < N) {
            const float x = to_f32<T>(W[r * N + c]);
            float dq;
            const uint8_t q = fp8_encode(x / s + 0.0f, 0, (fake & 0x100) ? 1 : 0, &dq);
            if (fake & 1) ((T*)out)[r * N + c] = from_f32<T>(opaque_f32(dq * s));
            else ((uint8_t*)out)[r * N + c] = q;
        }
    }
}

// eight codes per thread (N % 8 == 0, b % 8 == 0: one scale per run), 8-B loads, 16-B stores
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_block_dequant8(const uint8_t* __restrict__ W, const float* __restrict__ scales,
                                                            int64_t M, int64_t N, int b, int nbn, T* __restrict__ out) {
    const int64_t total8 = M * N / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        const int64_t e0 = i * 8, r = e0 / N, c = e0 - r * N;
        const float s = scales[(r / b) * nbn + c / b];
        const uint2 u = *reinterpret_cast<const uint2*>(W + e0);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = opaque_f32(fb_e4m3fn_to_f32((uint8_t)((e < 4 ? u.x >> (8 * e) : u.y >> (8 * (e - 4))) & 0xffu)) * s);
        store8<T>(out + e0, y);
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

// act_quant, b = 128 and a 16-B aligned tensor: 16 lanes per block, each lane one 16-B (16-bit types) load of its eight
// elements and one 8-B store of its codes; two blocks per lane group in flight
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_act_quant128(const T* __restrict__ X, int64_t nblocks, uint8_t* __restrict__ Y,
                                                          float* __restrict__ S) {
    const int sub = threadIdx.x & 15;
    const int64_t blk0 = (((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4) * 2;
    if (blk0 >= nblocks) return;
    const bool two = blk0 + 1 < nblocks;
    float v[2][8];
    load8<T>(X + blk0 * 128 + sub * 8, v[0]);
    if (two) load8<T>(X + (blk0 + 1) * 128 + sub * 8, v[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && !two) break;
        float am = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[h][e]));
        am = wave_max(am, 16);
        const float s = am / 448.0f;
        if (sub == 0) S[blk0 + h] = s;
        // plain divisor and numerators: the IEEE quotient from the refined reciprocal (quant_math.h); an all-zero block
        // (s = 0: the reference's 0 / 0) and anything unusual keep the division
        const bool plain = plain_pos(s) && am < 0x1p40f;
        const float yr = rcp_refined(s);
        uint8_t q[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const uint32_t c = plain ? fb_f32x2_to_e4m3fn(div_tail(v[h][e], s, yr), div_tail(v[h][e + 1], s, yr))
                                     : fb_f32x2_to_e4m3fn(v[h][e] / s, v[h][e + 1] / s);
            q[e] = (uint8_t)c;
            q[e + 1] = (uint8_t)(c >> 8);
        }
        store8_bytes(Y + (blk0 + h) * 128 + sub * 8, q);
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
//     acc = fma(partial * a_s[row], b_s, acc)    (kernel.py:226 with Triton's fp fusion, like k_fp8_block_gemm256 below)
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
                    acc[m][n][r] = __builtin_fmaf(part[m][n][r] * as, bs, acc[m][n][r]);   // as the Triton kernel compiles it
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


// ---------------------------------------------------------------------------------------------------------------
// k_fp8_block_gemm256 — the same product on the K = 64 fp8 MFMA (v_mfma_f32_32x32x64_f8f6f4, unit scales: twice the rate
// of the K = 16 form), for K % 128 == 0 and 16-B aligned operands below 4 GiB.
//   Workgroup tile 256 (m) x 256 (n), 8 waves as 2 (m) x 4 (n) = two per SIMD, wave tile 128 (m) x 64 (n), persistent grid
//   with one block of tiles per XCD and round (the CUs that share an L2 share operand panels).
//   Operands: one K block (128 B of every row) per stage, A panel + B panel = 64 KiB, two stages; staged by LDS-DMA in 1 KiB
//   pieces (8 rows x 128 B), the 16-B chunk index XOR-ed with ((row >> 1) & 7) on the DMA source address and on the read:
//   every ds_read_b128 lane group touches 16 distinct 16-B bank slots.
//   The MFMA runs transposed (D = B_tile . A_tile^T: n on the accumulator registers, m on the lanes), so a lane needs ONE
//   activation scale a_s[m, kb] per 32 x 32 accumulator and K block, and b_s[n / 128, kb] is wave-uniform (a scalar load).
//   Per K block and accumulator: part = 2 MFMAs from a zero C, then acc = fma(part * a_s, b_s, acc) on the fp32 VALU
//   (kernel.py:226 as Triton compiles it with fp fusion) — the second wave of the SIMD keeps the matrix pipe busy meanwhile.
//   Which k a lane's 32 operand bytes stand for does not matter as long as both operands use the same bytes: lane l takes
//   bytes 32 * (l >> 5) .. + 31 of its row's 64-B half, for A and for B.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int G2_T = 256;
static constexpr int G2_PANEL = G2_T * 128;        // 32 KiB
static constexpr int G2_STAGE = 2 * G2_PANEL;
static constexpr int G2_AS = 2 * G2_STAGE;         // a_s of one K block: 256 floats, two slots
static constexpr int G2_LDS = 2 * G2_STAGE + 2 * 1024;   // 130 KiB
static constexpr int G2_THREADS = 512;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Fp8GemmArgs {
    const uint8_t* A; const float* As; const uint8_t* B; const float* Bs;
    int64_t M, N, K;
    const void* bias; void* C;
    int ntm, ntn, sbm, sbn, nsn, nrounds;
    int fused;    // LLMC_FP8_GEMM_FUSED_SCALE: acc = fma(part, a_s * b_s, acc) — one VALU op per element and K block instead of the
                  // reference kernel's two (other rounding of the scale product: not bit-identical to the Triton kernel)
#ifdef LLMC_LAB
    int abl;      // tools/probes/fp8_gemm_lab.hip: 1 no accumulator update, 2 no DMA after a tile's first stage, 4 no MFMA, 16 / 32 update
                  // variants, 64 no barrier in the K loop, 128 no fragment reads after a tile's first
#endif
};

template <int DST>
__device__ __forceinline__ void g2_dma(i32x4 rsrc, uint32_t voff, uint32_t soff, uint32_t wvoff) {
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
                 :: "s"(soff), "v"(voff), "s"(rsrc), "s"(wvoff), "n"(DST) : "memory", "scc");
}
template <int DST>
__device__ __forceinline__ void g2_dma4(i32x4 rsrc, uint32_t voff, uint32_t soff, uint32_t m0base) {    // 4 B per lane
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %0 offen lds"
                 :: "s"(soff), "v"(voff), "s"(rsrc), "s"(m0base), "n"(DST) : "memory", "scc");
}
template <int I, int N, typename F> __device__ __forceinline__ void g2_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        g2_for<I + 1, N>(f);
    }
}

template <int DT, bool FUSED = false>     // FUSED: LLMC_FP8_GEMM_FUSED_SCALE (a template parameter: as a run-time branch it spilled 102 registers)
__global__ __launch_bounds__(G2_THREADS) void k_fp8_block_gemm256(Fp8GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];
    LDS_AS char* lds = (LDS_AS char*)g2_smem;
    if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // DMA destinations are immediates
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    const int nkb = (int)(a.K / 128);
    const uint32_t K32 = (uint32_t)a.K;

    i32x4 ra, rb;
    {
        const int64_t ab = a.M * a.K, bb = a.N * a.K;
        ra[0] = (int)(uint32_t)(uintptr_t)a.A;
        ra[1] = (int)((uint32_t)((uintptr_t)a.A >> 32) & 0xffffu);
        ra[2] = (int)(uint32_t)ab;
        ra[3] = 0x00020000;
        rb[0] = (int)(uint32_t)(uintptr_t)a.B;
        rb[1] = (int)((uint32_t)((uintptr_t)a.B >> 32) & 0xffffu);
        rb[2] = (int)(uint32_t)bb;
        rb[3] = 0x00020000;
    }
    // b_s [N / 128, nkb] fp32: read-only for the whole launch, read through the constant address space = scalar loads (which do
    // not ride on the vector memory counter the DMA is waited on with)
    typedef __attribute__((address_space(4))) const float cfloat_t;
    cfloat_t* bsc = (cfloat_t*)(uintptr_t)a.Bs;
    // a_s [M, nkb] fp32 through a buffer descriptor (rows past M read 0); the 256 scales of a K block go to LDS by DMA like the
    // operands (a compiler-visible vector load in the loop would make hipcc count vmcnt without the asm-issued DMA)
    i32x4 rs;
    rs[0] = (int)(uint32_t)(uintptr_t)a.As;
    rs[1] = (int)((uint32_t)((uintptr_t)a.As >> 32) & 0xffffu);
    rs[2] = (int)(uint32_t)(a.M * nkb * 4);
    rs[3] = 0x00020000;
    // DMA: wave wv moves pieces wv + 8 i (i = 0..3) of each panel; a piece = rows 8 p .. 8 p + 7, lane = (row lane >> 3, slot lane & 7)
    const uint32_t voff = (uint32_t)(lane >> 3) * K32 + (uint32_t)((((lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7))) << 4);
    const uint32_t wvoff = (uint32_t)wv * 1024u;
    // fragment reads: row (lane & 31) of a 32-row tile, chunks (4 kh + 2 h + e) ^ f
    const int f = ((lane & 31) >> 1) & 7, h = lane >> 5;
    int choff[2][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int e = 0; e < 2; ++e) choff[kh][e] = (lane & 31) * 128 + (((4 * kh + 2 * h + e) ^ f) << 4);
    const int baseA = wm * 128 * 128;                 // + j * 4096
    const int baseB = G2_PANEL + wn * 64 * 128;       // + i * 4096

    for (int round = 0; round < a.nrounds; ++round) {
        int tm, tn;
        {
            const int slot = blockIdx.x >> 3;
            const int si = slot / a.sbn, sj = slot - si * a.sbn;
            const int sid = round * 8 + (blockIdx.x & 7);
            const int sm = sid / a.nsn, sn = sid - sm * a.nsn;
            tm = sm * a.sbm + si;
            tn = sn * a.sbn + sj;
            if (tm >= a.ntm || tn >= a.ntn) continue;
        }
        const int64_t m0 = (int64_t)tm * G2_T, n0 = (int64_t)tn * G2_T;
        uint32_t sA[4], sB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sA[i] = (uint32_t)((m0 + 8 * (wv + 8 * i)) * a.K);
            sB[i] = (uint32_t)((n0 + 8 * (wv + 8 * i)) * a.K);
        }
        const uint32_t asoff = (uint32_t)((m0 + (wv & 3) * 64 + lane) * nkb * 4);     // DMA source of this lane (waves 0..3)
        const int asrd = G2_AS + (wm * 128 + (lane & 31)) * 4;                        // + j * 128 + slot * 1024: this lane's scale
        // stage `st` (byte offset of the ring slot: 0 or G2_STAGE) <- K block kb of both panels
        // piece P of this wave's nine: 0..7 operand pieces (A, B alternating), 8 = a_s of K block kb (rows 64 (wv & 3) .. + 63 of the
        // tile, slot kb & 1; waves 4..7 repeat what waves 0..3 move: no branch in the loop)
        auto dma_piece = [&](auto pc, uint32_t st, int kb) {
            constexpr int P = decltype(pc)::value;
            if constexpr (P == 8) g2_dma4<G2_AS>(rs, asoff, (uint32_t)kb * 4u, (uint32_t)((wv & 3) * 256 + (kb & 1) * 1024));
            else if constexpr ((P & 1) == 0) {
                g2_dma<(P >> 1) * 8192>(ra, voff, sA[P >> 1], wvoff + st);
                sA[P >> 1] += 128u;
            } else {
                g2_dma<G2_PANEL + (P >> 1) * 8192>(rb, voff, sB[P >> 1], wvoff + st);
                sB[P >> 1] += 128u;
            }
        };
        auto stage_dma = [&](uint32_t st, int kb) { g2_for<0, 9>([&](auto pc) { dma_piece(pc, st, kb); }); };
        // scales: a_s of this lane's four m rows (one per 32-row tile; rows past M read 0 through the descriptor's bound),
        // b_s of this wave's 128-column block
        const int nblk = __builtin_amdgcn_readfirstlane((int)((n0 + wn * 64) / 128));
        const int bsrow = ((int64_t)nblk * 128 < a.N ? nblk : 0) * nkb;

        f32x2 acc[2][4][8];       // pairs: the packed fp32 VALU forms the sums, the MFMAs never touch them
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[i][j][r] = f32x2{0.0f, 0.0f};

        stage_dma(0u, 0);

        // Fragment state carried from K block to K block: the first fragments of block k + 1 are read at the END of block k (after
        // its last MFMAs have issued and the barrier that publishes slot k + 1), under the two accumulator updates still pending,
        // so a block starts with its operands in registers instead of with a barrier, address arithmetic and an LDS round trip.
        i32x8 fb[2][2], fa[2][2];
        int adA[2][2];
        auto frag = [&](const int (&ad)[2][2], int imm, int kh) -> i32x8 {
            const i32x4 lo = *(LDS_AS const i32x4*)(lds + ad[kh][0] + imm);
            const i32x4 hi = *(LDS_AS const i32x4*)(lds + ad[kh][1] + imm);
            return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        // slot st has landed for every wave: read addresses (formed here, not hoisted: loop-invariant registers are scarce), both
        // n-tile operands and the first m-tile operand
        auto first_frags = [&](uint32_t st) {
            int adB[2][2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(adA[kh][e]) : "s"((int)st + baseA), "v"(choff[kh][e]));
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(adB[kh][e]) : "s"((int)st + baseB), "v"(choff[kh][e]));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) fb[i][kh] = frag(adB, i * 4096, kh);
            fa[0][0] = frag(adA, 0, 0);
            fa[0][1] = frag(adA, 0, 1);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        first_frags(0u);

        // One K block out of ring slot `st`: eight accumulators t = 2 j + i in a row (both n tiles' operand fragments stay in
        // registers, the m tile's are read once per K block), software-pipelined inside the wave: the MFMAs of t are issued, then
        // a DMA piece of the next K block, (t even) the fragment reads of the next m tile, then the fp32 update of accumulator
        // t - 1 runs under those MFMAs.
        auto kblock = [&](auto morec, uint32_t st, int kb) {
            constexpr bool more = decltype(morec)::value;        // another K block follows: request it
            float as_cur[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) as_cur[j] = *(LDS_AS const float*)(lds + asrd + (kb & 1) * 1024 + j * 128);
            const float bs = bsc[bsrow + kb];
            f32x16 pp[2];
            auto update = [&](auto tc) {
                constexpr int t = decltype(tc)::value, j = t >> 1, i = t & 1;
                // acc += part * a_s * b_s the way the reference's Triton kernel is compiled (fp fusion on: the second product
                // and the sum contract into one fma); scalar fp32 ops: packed ones cost twice the issue time
#ifdef LLMC_LAB
                if (a.abl & 1) return;
#endif
                if constexpr (FUSED) {   // opt-in: one op per element (the scale product rounded once per row and K block)
                    const float sc = as_cur[j] * bs;
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        acc[i][j][r] = f32x2{__builtin_fmaf(pp[t & 1][2 * r], sc, acc[i][j][r][0]), __builtin_fmaf(pp[t & 1][2 * r + 1], sc, acc[i][j][r][1])};
                    asm volatile("" : "+v"(acc[i][j][0]), "+v"(acc[i][j][1]), "+v"(acc[i][j][2]), "+v"(acc[i][j][3]),
                                 "+v"(acc[i][j][4]), "+v"(acc[i][j][5]), "+v"(acc[i][j][6]), "+v"(acc[i][j][7]));
                    return;
                }
#ifdef LLMC_LAB
                if (a.abl & 32) {        // the same op count, not reading the MFMA results
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float v0 = acc[i][j][r][1] * as_cur[j], v1 = acc[i][j][r][0] * as_cur[j];
                        acc[i][j][r] = f32x2{__builtin_fmaf(v0, bs, acc[i][j][r][0]), __builtin_fmaf(v1, bs, acc[i][j][r][1])};
                    }
                    asm volatile("" : "+v"(acc[i][j][0]), "+v"(acc[i][j][1]), "+v"(acc[i][j][2]), "+v"(acc[i][j][3]),
                                 "+v"(acc[i][j][4]), "+v"(acc[i][j][5]), "+v"(acc[i][j][6]), "+v"(acc[i][j][7]));
                    return;
                }
#endif
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float v0 = pp[t & 1][2 * r] * as_cur[j], v1 = pp[t & 1][2 * r + 1] * as_cur[j];
                    acc[i][j][r] = f32x2{__builtin_fmaf(v0, bs, acc[i][j][r][0]), __builtin_fmaf(v1, bs, acc[i][j][r][1])};
                }
                // the sums are formed HERE (an IR pass otherwise sinks them past the next barrier and keeps the partial
                // tiles alive)
                asm volatile("" : "+v"(acc[i][j][0]), "+v"(acc[i][j][1]), "+v"(acc[i][j][2]), "+v"(acc[i][j][3]),
                             "+v"(acc[i][j][4]), "+v"(acc[i][j][5]), "+v"(acc[i][j][6]), "+v"(acc[i][j][7]));
            };
            g2_for<0, 8>([&](auto tc) {
                constexpr int t = decltype(tc)::value, j = t >> 1, i = t & 1;
                f32x16 pz;
#pragma unroll
                for (int r = 0; r < 16; ++r) pz[r] = 0.0f;
#ifdef LLMC_LAB
                if (a.abl & 4) {
                    pp[t & 1] = pz;
                    pp[t & 1][0] = __builtin_bit_cast(float, fb[i][0][0] ^ fa[j & 1][0][1] ^ fb[i][1][2] ^ fa[j & 1][1][3]);
                } else
#endif
                {
                // the wave that has MFMAs to issue goes first: the other wave of the SIMD is in its update and can fill the gaps,
                // the matrix pipe cannot catch up on a late issue
                __builtin_amdgcn_s_setprio(3);
                pp[t & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[i][0], fa[j & 1][0], pz, 0, 0, 0, 0, 0, 0);
                pp[t & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[i][1], fa[j & 1][1], pp[t & 1], 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the next K block's nine pieces ride behind the MFMAs of the first three accumulators
                if constexpr (more) {
#ifdef LLMC_LAB
                    if (a.abl & 2) {
                    } else
#endif
                    {
                    if constexpr (t < 3) {
                        dma_piece(std::integral_constant<int, 3 * t>{}, st ^ (uint32_t)G2_STAGE, kb + 1);
                        dma_piece(std::integral_constant<int, 3 * t + 1>{}, st ^ (uint32_t)G2_STAGE, kb + 1);
                        dma_piece(std::integral_constant<int, 3 * t + 2>{}, st ^ (uint32_t)G2_STAGE, kb + 1);
                    }
                    }
                }
                if constexpr (i == 0 && j < 3) {
#ifdef LLMC_LAB
                    if (!(a.abl & 128))
#endif
                    {
                    fa[(j + 1) & 1][0] = frag(adA, (j + 1) * 4096, 0);
                    fa[(j + 1) & 1][1] = frag(adA, (j + 1) * 4096, 1);
                    }
                }
                if constexpr (more && t == 7) {
                    // every fragment of this slot is in registers: publish the next slot (my pieces landed, then everybody's)
                    // and fetch its first fragments while updates 6 and 7 are still to run
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LLMC_LAB
                    if (!(a.abl & 64))
#endif
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
#ifdef LLMC_LAB
                    if (!(a.abl & 128))
#endif
                    first_frags(st ^ (uint32_t)G2_STAGE);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (t > 0) update(std::integral_constant<int, t - 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            update(std::integral_constant<int, 7>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        uint32_t st = 0;
        for (int kb = 0; kb + 1 < nkb; ++kb) {
            kblock(std::true_type{}, st, kb);
            st ^= (uint32_t)G2_STAGE;
        }
        kblock(std::false_type{}, st, nkb - 1);
        // every wave is done with the last slot before the next tile's first DMA lands in slot 0 (nkb odd: the slot just read)
        __builtin_amdgcn_s_barrier();

        // epilogue: lane = row m, four consecutive n per register group (lanes 0-31 and 32-63 together: 16 B (8 B) runs of fp32
        // (16-bit) outputs per row and store)
        {
        // the output addresses are formed AFTER the K loop (opaque tile indices: hipcc otherwise computes them before the loop
        // and spills them across it)
        int tme = tm, tne = tn;
        asm volatile("" : "+s"(tme), "+s"(tne));
        const int64_t m0 = (int64_t)tme * G2_T, n0 = (int64_t)tne * G2_T;
        const bool full = m0 + G2_T <= a.M && n0 + G2_T <= a.N && (a.N & 7) == 0 && a.bias == nullptr;   // block-uniform
        if (full) {
            constexpr int ES = DT == LLMC_F32 ? 4 : 2;
            g2_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (DT == LLMC_F32) {
                    char* crow = (char*)a.C + ((m0 + wm * 128 + j * 32 + (lane & 31)) * a.N + n0 + wn * 64 + 4 * h) * ES;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2 v0 = acc[i][j][2 * q], v1 = acc[i][j][2 * q + 1];
                            *reinterpret_cast<float4*>(crow + (i * 32 + 8 * q) * ES) = float4{v0[0], v0[1], v1[0], v1[1]};
                        }
                } else {
                    // 16-bit outputs: a lane's run of four is 8 B. The two halves of the wave trade (v_permlane32_swap) the odd
                    // group of the lower lanes for the even group of the upper lanes: every lane then owns eight consecutive
                    // outputs = one 16-B store (half the store instructions: the epilogue is store-issue bound)
                    char* crow = (char*)a.C + ((m0 + wm * 128 + j * 32 + (lane & 31)) * a.N + n0 + wn * 64 + 8 * h) * ES;
                    auto pack = [&](f32x2 v) -> uint32_t {
                        if constexpr (DT == LLMC_BF16) return (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
                        else return (uint32_t)f32_to_f16_bits(v[0]) | ((uint32_t)f32_to_f16_bits(v[1]) << 16);
                    };
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int qp = 0; qp < 2; ++qp) {
                            const uint32_t x0 = pack(acc[i][j][4 * qp]), x1 = pack(acc[i][j][4 * qp + 1]);          // group 2 qp
                            const uint32_t y0 = pack(acc[i][j][4 * qp + 2]), y1 = pack(acc[i][j][4 * qp + 3]);      // group 2 qp + 1
                            const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                            // lanes 0-31: own x (n + 0..3), the partner's x (n + 4..7); lanes 32-63: the partner's y, own y
                            const uint4 ov = uint4{s0[0], s1[0], s0[1], s1[1]};
                            *reinterpret_cast<uint4*>(crow + (i * 32 + 16 * qp) * ES) = ov;
                        }
                }
            });
        } else {
            g2_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int64_t row = m0 + wm * 128 + j * 32 + (lane & 31);
                if (row < a.M) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int64_t col = n0 + wn * 64 + i * 32 + 8 * q + 4 * h + e;
                                if (col < a.N) {
                                    float y = rndc<DT>(acc[i][j][2 * q + (e >> 1)][e & 1]);
                                    if (a.bias) y = rndc<DT>(y + load_as_f32(a.bias, col, DT));
                                    store_from_f32(a.C, row * a.N + col, DT, y);
                                }
                            }
                }
            });
        }
        }
    }
}

static void g2_tile_order(Fp8GemmArgs& a, int grid) {
    const int spx = grid / 8;
    int best = 1;
    int64_t best_cost = -1;
    for (int sbn = 1; sbn <= 8; sbn *= 2) {
        if (spx % sbn) continue;
        const int sbm = spx / sbn;
        const int64_t padded = ceil_div64(a.ntm, sbm) * sbm * ceil_div64(a.ntn, sbn) * sbn;
        const int64_t cost = padded * 64 + (sbm + sbn);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sbn; }
    }
    a.sbn = best;
    a.sbm = spx / best;
    a.nsn = (int)ceil_div64(a.ntn, a.sbn);
    a.nrounds = (int)ceil_div64(ceil_div64(a.ntm, a.sbm) * a.nsn, 8);
}

}  // namespace llmc

using namespace llmc;

template <typename T>
static void launch_block_quant(const void* W, int64_t M, int64_t N, int block, float clamp_min, int z1, int fake, void* out,
                               float* scales, hipStream_t st) {
    const int nbn = (int)ceil_div64(N, block), nbm = (int)ceil_div64(M, block);
    const bool aligned = block == 128 && N % 8 == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0;
    if (!aligned) {
        hipLaunchKernelGGL((k_fp8_block_quant<T>), dim3(nbn, nbm), dim3(256), 0, st, (const T*)W, M, N, block, clamp_min, z1, fake, out, scales, nbn, 0);
        return;
    }
    const int fn = (int)(N / 128), fm = (int)(M / 128);          // whole blocks
    if (fn > 0 && fm > 0)
        hipLaunchKernelGGL((k_fp8_block_quant128<T>), dim3(fn, fm), dim3(256), 0, st, (const T*)W, M, N, clamp_min, z1, fake, out, scales, nbn);
    if (fn < nbn)     // ragged last block column, every block row
        hipLaunchKernelGGL((k_fp8_block_quant<T>), dim3(1, nbm), dim3(256), 0, st, (const T*)W, M, N, block, clamp_min, z1, fake, out, scales, nbn, 1);
    if (fm < nbm && fn > 0)     // ragged last block row, the whole block columns
        hipLaunchKernelGGL((k_fp8_block_quant<T>), dim3(fn, 1), dim3(256), 0, st, (const T*)W, M, N, block, clamp_min, z1, fake, out, scales, nbn, 2);
}

extern "C" int llmc_fp8_block_quant(const void* W, int dt, int64_t M, int64_t N, int block, float clamp_min,
                                    int fake, void* out, float* scales, llmc_stream_t stream) {
    LLMC_REQUIRE(W && out && scales && M > 0 && N > 0, "fp8_block_quant: null/empty argument");
    LLMC_REQUIRE(dtype_ok(dt), "fp8_block_quant: bad dtype");
    LLMC_REQUIRE(block >= 16 && block <= 128, "fp8_block_quant: block size must be in [16, 128]");
    hipStream_t st = (hipStream_t)stream;
    const int z1 = clamp_min > 0.0f ? 1 : 0;   // FloatQuantizer semantics replace a zero scale, kernel.py's do not
    switch (dt) {
        case LLMC_F16: launch_block_quant<f16_t>(W, M, N, block, clamp_min, z1, fake, out, scales, st); break;
        case LLMC_BF16: launch_block_quant<bf16_t>(W, M, N, block, clamp_min, z1, fake, out, scales, st); break;
        default: launch_block_quant<float>(W, M, N, block, clamp_min, z1, fake, out, scales, st); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

template <typename T>
static void launch_block_dequant(const void* W8, const float* scales, int64_t M, int64_t N, int block, void* out, hipStream_t st) {
    const int nbn = (int)ceil_div64(N, block);
    const bool vec = N % 8 == 0 && block % 8 == 0 && ((uintptr_t)W8 & 7) == 0 && ((uintptr_t)out & 15) == 0;
    const int64_t work = vec ? M * N / 8 : M * N;
    const int grid = (int)(ceil_div64(work, 256) < 65536 ? ceil_div64(work, 256) : 65536);
    if (vec) hipLaunchKernelGGL((k_fp8_block_dequant8<T>), dim3(grid), dim3(256), 0, st, (const uint8_t*)W8, scales, M, N, block, nbn, (T*)out);
    else hipLaunchKernelGGL((k_fp8_block_dequant<T>), dim3(grid), dim3(256), 0, st, (const uint8_t*)W8, scales, M, N, block, nbn, (T*)out);
}

extern "C" int llmc_fp8_block_dequant(const void* W8, const float* scales, int64_t M, int64_t N, int block, int out_dt,
                                      void* out, llmc_stream_t stream) {
    LLMC_REQUIRE(W8 && out && scales && M > 0 && N > 0, "fp8_block_dequant: null/empty argument");
    LLMC_REQUIRE(dtype_ok(out_dt) && block >= 1, "fp8_block_dequant: bad dtype / block");
    hipStream_t st = (hipStream_t)stream;
    switch (out_dt) {
        case LLMC_F16: launch_block_dequant<f16_t>(W8, scales, M, N, block, out, st); break;
        case LLMC_BF16: launch_block_dequant<bf16_t>(W8, scales, M, N, block, out, st); break;
        default: launch_block_dequant<float>(W8, scales, M, N, block, out, st); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

template <typename T>
static void launch_act_quant(const void* X, int64_t n_elem, int block, void* out8, float* scales, hipStream_t st) {
    const int64_t nblocks = n_elem / block;
    if (block == 128 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)out8 & 7) == 0) {
        const int grid = (int)ceil_div64(ceil_div64(nblocks, 2) * 16, 256);
        hipLaunchKernelGGL((k_fp8_act_quant128<T>), dim3(grid), dim3(256), 0, st, (const T*)X, nblocks, (uint8_t*)out8, scales);
    } else {
        const int grid = (int)ceil_div64(nblocks * 16, 256);
        hipLaunchKernelGGL((k_fp8_act_quant<T>), dim3(grid), dim3(256), 0, st, (const T*)X, nblocks, block, (uint8_t*)out8, scales);
    }
}

extern "C" int llmc_fp8_act_quant(const void* X, int dt, int64_t n_elem, int block, void* out8, float* scales,
                                  llmc_stream_t stream) {
    LLMC_REQUIRE(X && out8 && scales && n_elem > 0, "fp8_act_quant: null/empty argument");
    LLMC_REQUIRE(dtype_ok(dt), "fp8_act_quant: bad dtype");
    LLMC_REQUIRE(block >= 8 && block <= 128 && n_elem % block == 0, "fp8_act_quant: last dim must be a multiple of block <= 128");
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case LLMC_F16: launch_act_quant<f16_t>(X, n_elem, block, out8, scales, st); break;
        case LLMC_BF16: launch_act_quant<bf16_t>(X, n_elem, block, out8, scales, st); break;
        default: launch_act_quant<float>(X, n_elem, block, out8, scales, st); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_fp8_block_gemm(const void* A8, const float* a_s, const void* B8, const float* b_s, int64_t M,
                                   int64_t N, int64_t K, int out_dt, const void* bias, void* C, llmc_stream_t stream) {
    LLMC_REQUIRE(A8 && a_s && B8 && b_s && C && M > 0 && N > 0 && K > 0, "fp8_block_gemm: null/empty argument");
    const int fused = (out_dt & LLMC_FP8_GEMM_FUSED_SCALE) ? 1 : 0;
    out_dt &= ~LLMC_FP8_GEMM_FUSED_SCALE;
    LLMC_REQUIRE(out_dt == LLMC_F16 || out_dt == LLMC_BF16 || out_dt == LLMC_F32, "fp8_block_gemm: bad output dtype");
    hipStream_t st = (hipStream_t)stream;
    const int cus = device_cu_count() & ~7;
    const bool fast = K % 128 == 0 && M * K < (1ll << 32) && N * K < (1ll << 32) && (((uintptr_t)A8 | (uintptr_t)B8) & 15) == 0 &&
                      ((uintptr_t)C & 15) == 0 && cus >= 8 && M >= 128 && N >= 128;
    if (fast) {
        Fp8GemmArgs a;
        a.A = (const uint8_t*)A8; a.As = a_s; a.B = (const uint8_t*)B8; a.Bs = b_s;
        a.M = M; a.N = N; a.K = K; a.bias = bias; a.C = C;
        a.ntm = (int)ceil_div64(M, G2_T); a.ntn = (int)ceil_div64(N, G2_T);
        g2_tile_order(a, cus);
        a.fused = fused;
#ifdef LLMC_LAB
        a.abl = lab_env("LLMC_FP8_ABL") ? atoi(lab_env("LLMC_FP8_ABL")) : 0;
#endif
        const void* fn = out_dt == LLMC_F16 ? (fused ? (const void*)k_fp8_block_gemm256<LLMC_F16, true> : (const void*)k_fp8_block_gemm256<LLMC_F16>)
                       : out_dt == LLMC_BF16 ? (fused ? (const void*)k_fp8_block_gemm256<LLMC_BF16, true> : (const void*)k_fp8_block_gemm256<LLMC_BF16>)
                                             : (fused ? (const void*)k_fp8_block_gemm256<LLMC_F32, true> : (const void*)k_fp8_block_gemm256<LLMC_F32>);
        if (int rc = ensure_dynamic_lds(fn, G2_LDS)) return rc;
        void* kargs[] = {(void*)&a};
        LLMC_HIP_CHECK(hipLaunchKernel(fn, dim3(cus), dim3(G2_THREADS), kargs, (size_t)G2_LDS, st));
        return LLMC_OK;
    }
    dim3 grid((unsigned)ceil_div64(N, FG_T), (unsigned)ceil_div64(M, FG_T));
    switch (out_dt) {
        case LLMC_F16: hipLaunchKernelGGL((k_fp8_block_gemm<LLMC_F16>), grid, dim3(256), 0, st, (const uint8_t*)A8, a_s, (const uint8_t*)B8, b_s, M, N, K, bias, C); break;
        case LLMC_BF16: hipLaunchKernelGGL((k_fp8_block_gemm<LLMC_BF16>), grid, dim3(256), 0, st, (const uint8_t*)A8, a_s, (const uint8_t*)B8, b_s, M, N, K, bias, C); break;
        default: hipLaunchKernelGGL((k_fp8_block_gemm<LLMC_F32>), grid, dim3(256), 0, st, (const uint8_t*)A8, a_s, (const uint8_t*)B8, b_s, M, N, K, bias, C); break;
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
