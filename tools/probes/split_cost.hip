// Probe (MI355X): what the fp32 -> 3 x bf16 split of gemm3.hip costs per float4 on one wave per SIMD and on two, with and
// without the LDS stores, and the issue rate / dependent latency of v_cvt_pk_bf16_f32 next to v_pk_add_f32 and v_and_b32.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_cost tools/probes/split_cost.hip && /tmp/split_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))

template <int STORE>
__device__ __forceinline__ void split(f32x4 v, LDS_AS char* plane0, int off, uint32_t& sink) {
    f32x2_t a0 = {v[0], v[1]}, a1 = {v[2], v[3]};
    u32x2_t out[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x2_t h0 = __builtin_convertvector(a0, bf16x2_t);
        const bf16x2_t h1 = __builtin_convertvector(a1, bf16x2_t);
        out[t].x = __builtin_bit_cast(uint32_t, h0);
        out[t].y = __builtin_bit_cast(uint32_t, h1);
        if (t < 2) {
            a0 = a0 - __builtin_convertvector(h0, f32x2_t);
            a1 = a1 - __builtin_convertvector(h1, f32x2_t);
        }
    }
    if (STORE) {
#pragma unroll
        for (int t = 0; t < 3; ++t) *(LDS_AS u32x2_t*)(plane0 + t * 16384 + off) = out[t];
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) sink ^= out[t].x + out[t].y;
    }
}

template <int STORE, int NSPLIT>
__global__ __launch_bounds__(512) void k_split(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 16384];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int t = threadIdx.x & 255;
    f32x4 v[NSPLIT];
    for (int q = 0; q < NSPLIT; ++q) v[q] = f32x4{t * 0.37f + q, t * 1.1f - q, 3.3f * q + 1, t + 0.5f};
    uint32_t sink = 0;
    int off[NSPLIT];
    for (int q = 0; q < NSPLIT; ++q) { const int kr = (t >> 6) + 4 * (q & 7); off[q] = kr * 512 + ((((t & 63) >> 3) ^ (kr & 3)) << 6) + ((4 * (t & 63)) & 31) * 2; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NSPLIT; ++q) {
            split<STORE>(v[q], lds, off[q], sink);
            v[q][0] += 1.0f;      // a new value per iteration (nothing hoisted)
        }
        if (STORE) __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the stores of one step are done before the next
    }
    float r = (float)sink;
    for (int q = 0; q < NSPLIT; ++q) r += v[q][0];
    if (STORE) r += *(LDS_AS float*)(lds + 4 * t);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// OP: 0 v_cvt_pk_bf16_f32, 1 v_pk_add_f32, 2 v_and_b32, 3 v_lshlrev_b32;  DEP: every instruction reads the previous result
template <int OP, int DEP>
__global__ __launch_bounds__(512) void k_op(float* out, int iters) {
    float x[8], y = threadIdx.x * 0.5f + 1.0f;
    for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 0.25f + e;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            float& d = x[DEP ? 0 : (n & 7)];
            float& d2 = x[DEP ? 1 : ((n + 1) & 7)];
            if (OP == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(d) : "v"(y));
            if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[DEP ? 0 : 2 * (n & 3)]) : "v"(*(double*)&x[DEP ? 0 : 2 * ((n + 1) & 3)]));
            if (OP == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(d));
            if (OP == 3) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(d));
            (void)d2;
        }
    }
    float r = y;
    for (int e = 0; e < 8; ++e) r += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F>
static float time_it(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    for (int threads : {256, 512}) {
        float a = time_it([&] { k_split<0, 12><<<256, threads>>>(out, iters); });
        float b = time_it([&] { k_split<1, 12><<<256, threads>>>(out, iters); });
        printf("split, %d wave(s)/SIMD: registers only %.1f ns per float4 per wave, with the three LDS stores %.1f ns (12 per step: %.0f / %.0f ns)\n",
               threads / 256, a * 1e6 / iters / 12, b * 1e6 / iters / 12, a * 1e6 / iters, b * 1e6 / iters);
    }
    const char* names[4] = {"v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_and_b32", "v_lshlrev_b32"};
    float r[4][2];
    r[0][0] = time_it([&] { k_op<0, 0><<<256, 256>>>(out, iters); }); r[0][1] = time_it([&] { k_op<0, 1><<<256, 256>>>(out, iters); });
    r[1][0] = time_it([&] { k_op<1, 0><<<256, 256>>>(out, iters); }); r[1][1] = time_it([&] { k_op<1, 1><<<256, 256>>>(out, iters); });
    r[2][0] = time_it([&] { k_op<2, 0><<<256, 256>>>(out, iters); }); r[2][1] = time_it([&] { k_op<2, 1><<<256, 256>>>(out, iters); });
    r[3][0] = time_it([&] { k_op<3, 0><<<256, 256>>>(out, iters); }); r[3][1] = time_it([&] { k_op<3, 1><<<256, 256>>>(out, iters); });
    for (int o = 0; o < 4; ++o)
        printf("%-18s one wave per SIMD: independent %.2f ns per instruction, dependent chain %.2f ns\n", names[o], r[o][0] * 1e6 / iters / 32,
               r[o][1] * 1e6 / iters / 32);
    return 0;
}
