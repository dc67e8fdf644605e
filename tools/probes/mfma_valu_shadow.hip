// Probe (MI355X): how many independent fp32 VALU instructions issue in the shadow of a 64-cycle fp8 MFMA
// (v_mfma_f32_32x32x64_f8f6f4), one and two waves per SIMD. Per loop iteration: 2 MFMAs (two accumulators) + NV v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu_shadow tools/probes/mfma_valu_shadow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int PK, int NOMFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters, float s) {
    i32x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = 0x38383838 + threadIdx.x; b[e] = 0x30303030 + e; }
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; }
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = threadIdx.x * 0.001f + e;
    for (int it = 0; it < iters; ++it) {
        if (!NOMFMA) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&v[2 * (n % 8)]) : "v"(*(double*)&v[2 * ((n + 3) % 8)]));
            else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n % 16]) : "v"(s));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float t = 0;
    for (int r = 0; r < 16; ++r) t += c0[r] + c1[r] + v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NV, int PK, int NOMFMA>
static void run(float* out, int threads, const char* what) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NV, PK, NOMFMA><<<256, threads>>>(out, iters, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const int wps = threads / 256;
    printf("%-10s %d wave(s)/SIMD  NV=%2d %s: %.1f ns per iteration per wave-slot (2 MFMAs = %.1f ns at 2.4 GHz)\n", what, wps, NV,
           PK ? "pk_fma" : "fma", best * 1e6 / iters, 128 / 2.4);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
#define ROW(NV) run<NV, 0, 0>(out, 256, "mfma+valu"); run<NV, 0, 0>(out, 512, "mfma+valu");
    ROW(0) ROW(8) ROW(16) ROW(24) ROW(32) ROW(48) ROW(64)
    run<32, 0, 1>(out, 256, "valu only"); run<32, 0, 1>(out, 512, "valu only");
    run<16, 1, 0>(out, 256, "mfma+valu"); run<16, 1, 0>(out, 512, "mfma+valu");
    run<16, 1, 1>(out, 256, "valu only"); run<16, 1, 1>(out, 512, "valu only");
    return 0;
}
