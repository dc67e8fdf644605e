// Probe (MI355X): semantics and rate of the gfx950 K=64 fp8 MFMA, and the hardware fp32 -> e4m3 conversion.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp8_probe tools/probes/fp8_mfma_probe.hip && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static float e4m3_to_f32(uint8_t v) {
    const uint32_t e = (v >> 3) & 0xf, m = v & 7;
    float r;
    if ((v & 0x7f) == 0x7f) return NAN;
    if (e == 0) r = (float)m * 0.001953125f;
    else { uint32_t b = ((e + 120u) << 23) | (m << 20); memcpy(&r, &b, 4); }
    return (v & 0x80) ? -r : r;
}

template <int MODE>
__global__ void k_one(const uint8_t* A, const uint8_t* B, float* D) {
    // A [32][64], B [32][64] (row = i / j, k contiguous); lane l: row l & 31, bytes 32 * (l >> 5) .. + 31
    const int l = threadIdx.x;
    i32x8 a = *(const i32x8*)(A + (l & 31) * 64 + 32 * (l >> 5));
    i32x8 b = *(const i32x8*)(B + (l & 31) * 64 + 32 * (l >> 5));
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    if (MODE == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
        D[i * 32 + j] = c[r];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    i32x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = 0x38383838 + threadIdx.x; b[e] = 0x30303030 + e; }
    f32x16 c[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (MODE == 0) c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[t], 0, 0, 0, 0, 0, 0);
            else if (MODE == 1) c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else {
                long la = ((long)a[0] << 32) | (unsigned)a[1], lb = ((long)b[0] << 32) | (unsigned)b[1];
                c[t] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c[t], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void k_cvt(const float* x, uint8_t* y, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (2 * i + 1 < n) {
        const int p = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        y[2 * i] = p & 0xff;
        y[2 * i + 1] = (p >> 8) & 0xff;
    }
}

static uint8_t sw_f32_to_e4m3fn(float x) {
    uint32_t b; memcpy(&b, &x, 4);
    const uint32_t sign = (b >> 24) & 0x80u, ab = b & 0x7fffffffu;
    if (ab > 0x7f800000u) return (uint8_t)(sign | 0x7f);
    float ax; memcpy(&ax, &ab, 4);
    if (ax < 0.015625f) return (uint8_t)(sign | (uint32_t)rintf(ax * 512.0f));
    uint32_t r = ab + 0x7ffffu + ((ab >> 20) & 1u);
    r &= 0xfff00000u;
    if (r > 0x43e00000u) return (uint8_t)(sign | 0x7f);
    return (uint8_t)(sign | (((r >> 23) - 120u) << 3) | ((r >> 20) & 7u));
}

int main() {
    // 1. semantics
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& v : A) { do v = rnd() & 0xff; while ((v & 0x7f) == 0x7f); }
    for (auto& v : B) { do v = rnd() & 0xff; while ((v & 0x7f) == 0x7f); }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) k_one<0><<<1, 64>>>(dA, dB, dD); else k_one<1><<<1, 64>>>(dA, dB, dD);
        std::vector<float> D(1024);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double maxrel = 0; int exact = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < 64; ++k) {
                const double t = (double)e4m3_to_f32(A[i * 64 + k]) * (double)e4m3_to_f32(B[j * 64 + k]);
                ref += t;
                mag += fabs(t);
            }
            const double d = fabs(D[i * 32 + j] - ref) / (mag + 1e-30);
            if (d > maxrel) maxrel = d;
            exact += (float)ref == D[i * 32 + j];
        }
        printf("mfma 32x32x64 fp8 %s: max |err| / sum |terms| vs fp64 dot %.3g, %d / 1024 equal to the rounded exact sum\n",
               mode ? "scale 0x7f" : "non-scaled", maxrel, exact);
    }
    // 2. rate
    float* dout; hipMalloc(&dout, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_rate<0><<<256, 256>>>(dout, iters); else if (mode == 1) k_rate<1><<<256, 256>>>(dout, iters); else k_rate<2><<<256, 256>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double kk = mode == 2 ? 16 : 64;
        const double fl = 2.0 * 32 * 32 * kk * 8 * iters * 1024.0;
        printf("rate %s: %.1f TFLOP/s (1 wave per SIMD, 8 independent accumulators)\n",
               mode == 0 ? "32x32x64 f8f6f4 non-scaled" : mode == 1 ? "32x32x64 f8f6f4 scaled" : "32x32x16 fp8_fp8", fl / (ms * 1e-3) / 1e12);
    }
    // 3. hardware conversion
    const int n = 1 << 22;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) {
        uint32_t b;
        if (i < (1 << 21)) b = (rnd() << 8) ^ rnd();              // random bit patterns
        else { float f = ldexpf((float)(rnd() & 0xffffff) / 16777216.0f * 2.0f - 1.0f, (int)(rnd() % 20) - 9); memcpy(&b, &f, 4); }
        memcpy(&x[i], &b, 4);
    }
    // exact ties and range edges
    float edges[] = {448.0f, 464.0f, 463.99f, 464.01f, 480.0f, 1e9f, INFINITY, -INFINITY, 0.015625f, 0.0146484375f, 0.0009765625f, 0.00097656f, 0.0029296875f, -0.0f, 0.0f, 17.0f, 18.0f, 19.0f, 1.0625f, 1.1875f};
    for (size_t i = 0; i < sizeof(edges) / 4; ++i) x[i] = edges[i];
    float* dx; uint8_t* dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k_cvt<<<n / 512, 256>>>(dx, dy, n);
    std::vector<uint8_t> y(n);
    hipMemcpy(y.data(), dy, n, hipMemcpyDeviceToHost);
    long bad = 0, bad_in = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        const uint8_t w = sw_f32_to_e4m3fn(x[i]);
        if (w != y[i]) {
            ++bad;
            const bool inr = fabsf(x[i]) <= 464.0f;     // finite, representable after rounding
            bad_in += inr;
            if (shown < 12 && (inr || shown < 4)) { printf("  cvt differs: x=%g (0x%08x) hw=0x%02x torch=0x%02x\n", x[i], *(uint32_t*)&x[i], y[i], w); ++shown; }
        }
    }
    printf("v_cvt_pk_fp8_f32 vs torch.float8_e4m3fn cast: %ld / %d differ, %ld of them with |x| <= 464\n", bad, n, bad_in);
    return 0;
}
