// Probe: what v_mfma_f32_32x32x2_f32 sustains on gfx950 with nothing else in the loop, with the operand reads of k_sgemm's
// inner loop, and with 1 / 2 / 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o p && ./p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, int MODE>   // MODE 0: registers only; 1: operands re-read from LDS every k-pair (k_sgemm's pattern); 2: 16x16x4 registers only
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float seed) {
    __shared__ float lds[2 * 16 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 16 * 132; i += 256) lds[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x16 acc[NACC];
    f32x4 acc4[NACC * 4];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int n = 0; n < NACC * 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[n][r] = 0.f;
    float fa = seed * lane, fb = seed + lane;
    const float* pa = lds + (lane >> 5) * 132 + (lane & 31);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float a0 = pa[2 * kk * 132], a1 = pa[2 * kk * 132 + 32], b0 = pa[16 * 132 + 2 * kk * 132], b1 = pa[16 * 132 + 2 * kk * 132 + 32];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1 % NACC], 0, 0, 0);
                acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2 % NACC], 0, 0, 0);
                acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3 % NACC], 0, 0, 0);
            }
        } else if (MODE == 0) {
#pragma unroll
            for (int kk = 0; kk < 32 / NACC; ++kk)
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[n], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 16 / NACC; ++kk)
#pragma unroll
                for (int n = 0; n < NACC * 4; ++n) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc4[n], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[n][r];
#pragma unroll
    for (int n = 0; n < NACC * 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc4[n][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC, int MODE>
static void run(const char* what, int wg_per_cu, float* out) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = cus * wg_per_cu;
    k_probe<NACC, MODE><<<grid, 256>>>(out, 100, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k_probe<NACC, MODE><<<grid, 256>>>(out, iters, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per iteration and wave: 32 MFMAs of 32x32x2 (4096 flop) or 64 of 16x16x4 (2048 flop... 16*16*4*2)
    const double flop = 5.0 * grid * 4.0 * iters * 32 * 4096.0;
    printf("%-46s acc=%d wg/CU=%d grid=%d: %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", what, NACC, wg_per_cu, grid, ms / 5, flop / (ms * 1e-3) / 1e12,
           flop / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    for (int w : {1, 2, 4}) {
        run<4, 0>("32x32x2 f32, registers only", w, out);
        run<1, 0>("32x32x2 f32, registers only, ONE accumulator", w, out);
        run<2, 0>("32x32x2 f32, registers only, two accumulators", w, out);
        run<4, 1>("32x32x2 f32, operands from LDS each pair", w, out);
        run<4, 2>("16x16x4 f32, registers only (16 accumulators)", w, out);
    }
    return 0;
}
