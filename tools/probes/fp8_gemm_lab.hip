// Lab (MI355X): the K = 64 block-scaled fp8 GEMM with parts of its K-block body switched off (LLMC_LAB build of the product source).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DLLMC_LAB -Iinclude -Illmc_amd/csrc \
//       -o tools/probes/fp8_gemm_lab tools/probes/fp8_gemm_lab.hip llmc_amd/csrc/abi.hip
#include "../../llmc_amd/csrc/fp8_block.hip"
#include <cstdio>
#include <vector>

int main() {
    const int64_t M = 16384;
    const int64_t shapes[][2] = {{14336, 4096}, {4096, 14336}};
    const char* names[] = {"full", "no accumulator update", "no DMA after a tile's first stage", "no update, no DMA", "no barrier in the K loop",
                           "no fragment reads after the first", "no barrier, no reads", "no update, no DMA, no barrier", "no update, no DMA, no reads",
                           "no update, no DMA, no barrier, no reads (MFMA stream + loop)"};
    const int abls[] = {0, 1, 2, 3, 64, 128, 192, 67, 131, 195};
    for (auto& sh : shapes) {
        const int64_t N = sh[0], K = sh[1], nkb = K / 128;
        uint8_t *A, *B; float *As, *Bs; void* C;
        hipMalloc(&A, M * K); hipMalloc(&B, N * K); hipMalloc(&As, M * nkb * 4); hipMalloc(&Bs, (N / 128) * nkb * 4); hipMalloc(&C, M * N * 2);
        std::vector<uint8_t> h(M * K > N * K ? M * K : N * K);
        uint32_t s = 1;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 24) & 0x77; }      // finite e4m3 values
        hipMemcpy(A, h.data(), M * K, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), N * K, hipMemcpyHostToDevice);
        std::vector<float> f(M * nkb, 1.0f);
        hipMemcpy(As, f.data(), M * nkb * 4, hipMemcpyHostToDevice); hipMemcpy(Bs, f.data(), (N / 128) * nkb * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (size_t v = 0; v < sizeof(abls) / sizeof(int); ++v) {
            char buf[16]; snprintf(buf, sizeof buf, "%d", abls[v]); setenv("LLMC_FP8_ABL", buf, 1);
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                llmc_fp8_block_gemm(A, As, B, Bs, M, N, K, LLMC_BF16, nullptr, C, nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("N=%ld K=%ld %-48s %.3f ms = %.0f TFLOP/s-equivalent\n", (long)N, (long)K, names[v], best, 2.0 * M * N * K / (best * 1e-3) / 1e12);
        }
        hipFree(A); hipFree(B); hipFree(As); hipFree(Bs); hipFree(C);
    }
    return 0;
}
