// Probe (MI355X): sustained rate of a bare bf16 MFMA stream on TOGGLING operands (random bit patterns that change from one
// MFMA to the next: the power-limited case of profiles/NOTES.md "MFMA stream only, operands alternating between random
// register sets" = 0.645 of peak for v_mfma_f32_32x32x16_bf16) for the two dense bf16 shapes of gfx950:
//   v_mfma_f32_32x32x16_bf16 (16 accumulators of 32x32 per wave, what k_syrk4 issues) and
//   v_mfma_f32_16x16x32_bf16 (64 accumulators of 16x16: the shape the library GEMM names in its kernel, MI16x16x1),
// one wave per SIMD (256 threads per workgroup, one workgroup per CU), ~40 ms per measurement so that the clock settles.
// Also the same two streams on CONSTANT operands (no toggling) for the clock-unconstrained ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_shape_power tools/probes/mfma_shape_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x8 rnd_frag(unsigned seed) {
    s16x8 w;
    for (int e = 0; e < 8; ++e) {
        unsigned h = (threadIdx.x * 977u + e * 131u + seed * 7919u + 12345u) * 2654435761u;
        w[e] = (short)((h >> 16 & 0x83ff) | 0x3c00 | ((h >> 3) & 0x0300));     // random sign / mantissa, exponents near 1
    }
    return w;
}

// SHAPE 32: 4 A x 4 B fragments, 16 accumulators (256 registers), 16 MFMAs per slice; two slices alternate (toggle)
template <int TOGGLE>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    s16x8 a0[4], b0[4], a1[4], b1[4];
    for (int i = 0; i < 4; ++i) {
        a0[i] = rnd_frag(i); b0[i] = rnd_frag(10 + i);
        a1[i] = TOGGLE ? rnd_frag(20 + i) : a0[i]; b1[i] = TOGGLE ? rnd_frag(30 + i) : b0[i];
        asm volatile("" : "+v"(a0[i]), "+v"(b0[i]), "+v"(a1[i]), "+v"(b1[i]));
    }
    f32x16 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mi = i >> 2, ni = (mi & 1) ? 3 - (i & 3) : (i & 3);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[mi]), __builtin_bit_cast(bf16x8, b0[ni]), acc[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mi = i >> 2, ni = (mi & 1) ? 3 - (i & 3) : (i & 3);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[mi]), __builtin_bit_cast(bf16x8, b1[ni]), acc[mi][ni], 0, 0, 0);
        }
    }
    float t = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) t += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

// SHAPE 16: 8 A x 8 B fragments (16 rows x 32 k each), 64 accumulators of 4 registers (256 registers), 64 MFMAs per slice of 32 k
template <int TOGGLE>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    s16x8 a0[8], b0[8], a1[8], b1[8];
    for (int i = 0; i < 8; ++i) {
        a0[i] = rnd_frag(i); b0[i] = rnd_frag(10 + i);
        a1[i] = TOGGLE ? rnd_frag(20 + i) : a0[i]; b1[i] = TOGGLE ? rnd_frag(30 + i) : b0[i];
        asm volatile("" : "+v"(a0[i]), "+v"(b0[i]), "+v"(a1[i]), "+v"(b1[i]));
    }
    f32x4 acc[8][8];
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int mi = i >> 3, ni = (mi & 1) ? 7 - (i & 7) : (i & 7);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0[mi]), __builtin_bit_cast(bf16x8, b0[ni]), acc[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int mi = i >> 3, ni = (mi & 1) ? 7 - (i & 7) : (i & 7);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[mi]), __builtin_bit_cast(bf16x8, b1[ni]), acc[mi][ni], 0, 0, 0);
        }
    }
    float t = 0;
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) t += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <typename F>
static void run(const char* what, F launch, double flop_per_iter_per_wave, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float last = 0;
    for (int rep = 0; rep < 4; ++rep) {      // back to back: the later repetitions run at the sustained clock
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        const double tf = flop_per_iter_per_wave * iters * 1024.0 / (last * 1e-3) / 1e12;
        printf("%-44s rep %d: %7.2f ms  %7.0f TFLOP/s = %.3f of 2.5 PF\n", what, rep, last, tf, tf / 2500.0);
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    int ncu = 256;
    const double f32 = 32.0 * 2 * 32 * 32 * 16, f16 = 128.0 * 2 * 16 * 16 * 32;   // flop per loop iteration per wave
    const int it32 = 60000, it16 = 60000;    // 32 x 32-cycle and 128 x 8-cycle... both 1024 matrix-pipe cycles per iteration -> ~26-40 ms
    run("32x32x16 bf16, toggling operands", [&](int it) { k32<1><<<ncu, 256>>>(out, it); }, f32, it32);
    run("16x16x32 bf16, toggling operands", [&](int it) { k16<1><<<ncu, 256>>>(out, it); }, f16, it16);
    run("32x32x16 bf16, constant operands", [&](int it) { k32<0><<<ncu, 256>>>(out, it); }, f32, it32);
    run("16x16x32 bf16, constant operands", [&](int it) { k16<0><<<ncu, 256>>>(out, it); }, f16, it16);
    run("32x32x16 bf16, toggling operands (again)", [&](int it) { k32<1><<<ncu, 256>>>(out, it); }, f32, it32);
    run("16x16x32 bf16, toggling operands (again)", [&](int it) { k16<1><<<ncu, 256>>>(out, it); }, f16, it16);
    return 0;
}
