// Hardware probe (run on MI355X): pins the lane maps this repo's MFMA kernels rely on.
//   H1  ds_read_b64_tr_b16: lane l (group G=l>>4, p=l&15) with per-lane byte address A(l) receives
//       elem j = 16-bit word at A(16G + 4j + p/4) + 2*(p%4)   (a 4x16 transpose inside each 16-lane group)
//   H2  v_mfma_f32_32x32x16_bf16: A lane l = A[i=l&31][k=8*(l>>5)+e]; B lane l = B[k=8*(l>>5)+e][j=l&31];
//       C reg r of lane l = C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]
//   H3  v_mfma_f32_16x16x32_bf16: A lane l = A[l&15][8*(l>>4)+e]; B = B[8*(l>>4)+e][l&15];
//       C reg r = C[4*(l>>4)+r][l&15]
//   H4  v_mfma_f32_32x32x2_f32 / 16x16x4_f32 operand maps (guide §3)
// Build: hipcc --offload-arch=gfx950 -O2 probe_mfma_tr16.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T, p) reinterpret_cast<__attribute__((address_space(3))) T*>((__attribute__((address_space(3))) char*)(p))

__global__ void k_tr16(const uint16_t* in, const int* slot, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = in[i];
    __syncthreads();
    bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4_t, (char*)lds + slot[l] * 8));
    uint16_t r[4];
    __builtin_memcpy(r, &a, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}

__device__ __bf16 mk(float f) { return (__bf16)f; }

__global__ void k_mfma32(const float* A, const float* B, float* C) {  // A[32][16], B[16][32]
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = mk(A[(l & 31) * 16 + 8 * (l >> 5) + e]);
        b[e] = mk(B[(8 * (l >> 5) + e) * 32 + (l & 31)]);
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_mfma16(const float* A, const float* B, float* C) {  // A[16][32], B[32][16]
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = mk(A[(l & 15) * 32 + 8 * (l >> 4) + e]);
        b[e] = mk(B[(8 * (l >> 4) + e) * 16 + (l & 15)]);
    }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_mfma32f(const float* A, const float* B, float* C) {  // A[32][2], B[2][32]
    int l = threadIdx.x;
    float a = A[(l & 31) * 2 + (l >> 5)], b = B[(l >> 5) * 32 + (l & 31)];
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_mfma16f(const float* A, const float* B, float* C) {  // A[16][4], B[4][16]
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

static int check_mm(const char* name, const std::vector<float>& A, const std::vector<float>& B,
                    const std::vector<float>& C, int M, int N, int K) {
    int bad = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            float ref = 0;
            for (int k = 0; k < K; ++k) ref += A[i * K + k] * B[k * N + j];
            if (ref != C[i * N + j]) ++bad;
        }
    printf("%s: %s (%d mismatches)\n", name, bad ? "FAIL" : "PASS", bad);
    return bad;
}

template <typename F> static void run_mm(const char* name, F kern, int M, int N, int K, int& fails) {
    std::vector<float> A(M * K), B(K * N), C(M * N, -1);
    for (auto& v : A) v = (float)(rand() % 15 - 7);
    for (auto& v : B) v = (float)(rand() % 13 - 6);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    fails += check_mm(name, A, B, C, M, N, K) != 0;
    hipFree(dA); hipFree(dB); hipFree(dC);
}

int main() {
    int fails = 0;
    srand(7);
    {   // H1
        std::vector<uint16_t> in(1024), out(256);
        for (int i = 0; i < 1024; ++i) in[i] = (uint16_t)(i * 7 + 3);
        std::vector<int> slot(64);
        for (int trial = 0; trial < 3; ++trial) {
            for (int l = 0; l < 64; ++l) slot[l] = trial == 0 ? l : (trial == 1 ? (l * 37 + 5) % 128 : rand() % 128);
            uint16_t *di, *dout; int* ds;
            hipMalloc(&di, 2048); hipMalloc(&dout, 512); hipMalloc(&ds, 256);
            hipMemcpy(di, in.data(), 2048, hipMemcpyHostToDevice);
            hipMemcpy(ds, slot.data(), 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_tr16, dim3(1), dim3(64), 0, 0, di, ds, dout);
            hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    int G = l >> 4, p = l & 15;
                    int src_lane = 16 * G + 4 * j + p / 4;
                    uint16_t ref = in[slot[src_lane] * 4 + (p % 4)];
                    if (ref != out[l * 4 + j]) ++bad;
                }
            printf("H1 tr16 trial %d: %s (%d mismatches)\n", trial, bad ? "FAIL" : "PASS", bad);
            if (bad && trial == 0) {
                for (int l = 0; l < 64; ++l) {
                    printf("  lane %2d:", l);
                    for (int j = 0; j < 4; ++j) printf(" %5d", (out[l * 4 + j] - 3) / 7);
                    printf("\n");
                }
            }
            fails += bad != 0;
            hipFree(di); hipFree(dout); hipFree(ds);
        }
    }
    run_mm("H2 mfma_f32_32x32x16_bf16", k_mfma32, 32, 32, 16, fails);
    run_mm("H3 mfma_f32_16x16x32_bf16", k_mfma16, 16, 16, 32, fails);
    run_mm("H4a mfma_f32_32x32x2_f32", k_mfma32f, 32, 32, 2, fails);
    run_mm("H4b mfma_f32_16x16x4_f32", k_mfma16f, 16, 16, 4, fails);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device: %s, CUs %d, clock %d kHz, LDS/block %zu\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.sharedMemPerBlock);
    printf("probe %s\n", fails ? "FAILED" : "OK");
    return fails ? 1 : 0;
}
