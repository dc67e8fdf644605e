// Phase timing of k_potrf_inv (one workgroup factoring + inverting a 128x128 diagonal block) on MI355X.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//     -DLLMC_PROBE_STAMPS -Iinclude -Illmc_amd/csrc tools/probes/probe_potrf.hip llmc_amd/csrc/sgemm.hip \
//     llmc_amd/csrc/abi.hip llmc_amd/csrc/gemm3.hip -o tools/probes/probe_potrf
#include "../../llmc_amd/csrc/cholesky.hip"
#include <stdio.h>
#include <math.h>
#include <vector>

int main() {
    const int n = 128, ld = 128;
    std::vector<float> h(n * ld);
    // SPD: A = B B^T + n I with a fixed pseudo-random B
    std::vector<float> b(n * n);
    unsigned s = 12345;
    for (auto& x : b) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            float acc = (i == j) ? (float)n : 0.0f;
            for (int k = 0; k < n; ++k) acc += b[i * n + k] * b[j * n + k];
            h[i * ld + j] = acc;
        }
    float *W, *V; int* info;
    hipMalloc(&W, h.size() * 4); hipMalloc(&V, n * n * 4); hipMalloc(&info, 4);
    hipMemset(info, 0, 4);
    const size_t lds = (llmc::NB * llmc::PLD + 32 * llmc::PLD + 64) * sizeof(float);
    hipFuncSetAttribute((const void*)llmc::k_potrf_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long long st[32];
    for (int it = 0; it < 3; ++it) {
        hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(llmc::k_potrf_inv, dim3(1), dim3(256), lds, 0, W, (int64_t)ld, 0, n, V, info);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(llmc::g_potrf_stamps), sizeof(st));
        printf("run %d: event %.1f us; stamps (clock64 ticks from kernel start):\n", it, ms * 1e3);
        const char* nm[13] = {"start", "loaded", "potrf32[0]", "panel+trail[0]", "potrf32[1]", "panel+trail[1]",
                              "potrf32[2]", "panel+trail[2]", "potrf32[3]", "panel+trail[3]", "U stored(issue)",
                              "inverse levels", "V stored(issue)"};
        for (int i = 1; i <= 12; ++i) printf("  %-18s +%lld (total %lld)\n", nm[i], st[i] - st[i - 1], st[i] - st[0]);
    }
    // correctness: U^T U = A, V U = I (host fp64), worst 32x32 block reported
    std::vector<float> hu(n * ld), hv(n * n);
    hipMemcpy(hu.data(), W, hu.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hv.data(), V, hv.size() * 4, hipMemcpyDeviceToHost);
    double eu = 0, ev = 0; int bu = -1, bv = -1;
    double bue[16] = {0}, bve[16] = {0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < n; ++k) {
                s1 += (double)hu[k * ld + i] * hu[k * ld + j];
                s2 += (double)hv[i * n + k] * hu[k * ld + j];
            }
            double d1 = fabs(s1 - h[i * ld + j]) / n, d2 = fabs(s2 - (i == j ? 1.0 : 0.0));
            { int bb = (i / 32) * 4 + j / 32; if (d1 > bue[bb]) bue[bb] = d1; if (d2 > bve[bb]) bve[bb] = d2; }
            if (d1 > eu) { eu = d1; bu = (i / 32) * 4 + j / 32; }
            if (d2 > ev) { ev = d2; bv = (i / 32) * 4 + j / 32; }
        }
    printf("max |U^T U - A|/n = %.3g (block %d)   max |V U - I| = %.3g (block %d)\n", eu, bu, ev, bv);
    for (int r = 0; r < 4; ++r) {
        printf("  U err row %d: %.2e %.2e %.2e %.2e   V err: %.2e %.2e %.2e %.2e\n", r, bue[4 * r], bue[4 * r + 1],
               bue[4 * r + 2], bue[4 * r + 3], bve[4 * r], bve[4 * r + 1], bve[4 * r + 2], bve[4 * r + 3]);
    }
    {   // leading 6x6 of V against the host inverse of the leading 6x6 of U (upper triangular: exact sub-inverse)
        double ui[6][6] = {{0}};
        for (int j = 0; j < 6; ++j)
            for (int i = j; i >= 0; --i) {
                double s = (i == j) ? 1.0 : 0.0;
                for (int k = i + 1; k <= j; ++k) s -= (double)hu[i * ld + k] * ui[k][j];
                ui[i][j] = s / hu[i * ld + i];
            }
        for (int i = 0; i < 6; ++i) {
            printf("  V[%d][0:6] =", i);
            for (int j = 0; j < 6; ++j) printf(" % .5e", hv[i * n + j]);
            printf("\n  ref       =");
            for (int j = 0; j < 6; ++j) printf(" % .5e", ui[i][j]);
            printf("\n");
        }
    }
    {   // full host inverse of U; worst entries of the leading 32x32 block
        std::vector<double> ui(n * n, 0.0);
        for (int j = 0; j < n; ++j)
            for (int i = j; i >= 0; --i) {
                double s = (i == j) ? 1.0 : 0.0;
                for (int k = i + 1; k <= j; ++k) s -= (double)hu[i * ld + k] * ui[k * n + j];
                ui[i * n + j] = s / hu[i * ld + i];
            }
        int cnt = 0;
        for (int i = 0; i < 32 && cnt < 12; ++i)
            for (int j = 0; j < 32 && cnt < 12; ++j)
                if (fabs(hv[i * n + j] - ui[i * n + j]) > 1e-6) {
                    printf("  V[%d][%d] = % .5e ref % .5e\n", i, j, hv[i * n + j], ui[i * n + j]);
                    ++cnt;
                }
    }
    int hi; hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost);
    printf("info=%d\n", hi);
    return 0;
}
