// Does a CU-masked stream (hipExtStreamCreateWithCUMask) keep compute units free for kernels of another stream?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o tools/probes/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
#include <chrono>
#include <thread>

__global__ void k_where(unsigned* out, long long spin_ticks) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = clock64();
    while (clock64() - t0 < spin_ticks) { __builtin_amdgcn_s_sleep(4); }
}
__global__ void k_small(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1; }

static int distinct_cus(const std::vector<unsigned>& v, int n) {
    std::set<unsigned long long> s;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        s.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    return (int)s.size();
}

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int words = (ncu + 31) / 32;
    uint32_t mask[32];
    for (int w = 0; w < 32; ++w) mask[w] = 0xffffffffu;
    for (int k = 0; k < 8; ++k)
        for (int o = 0; o <= 8; o += 8) { int bit = 33 * k + o; if (bit < ncu) mask[bit >> 5] &= ~(1u << (bit & 31)); }
    hipStream_t sm = nullptr, sn = nullptr, sp = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&sm, words, mask);
    printf("CUs %d, mask words %d, hipExtStreamCreateWithCUMask -> %d (%s)\n", ncu, words, (int)e, hipGetErrorString(e));
    uint32_t back[32] = {0};
    if (e == hipSuccess) {
        hipError_t e2 = hipExtStreamGetCUMask(sm, words, back);
        printf("read back (%d):", (int)e2);
        for (int w = 0; w < words; ++w) printf(" %08x", back[w]);
        printf("\n");
    }
    hipStreamCreateWithFlags(&sn, hipStreamNonBlocking);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi);
    const int NB = 4096;
    unsigned* d; hipMalloc(&d, NB * 8);
    std::vector<unsigned> h(NB * 2);
    // which CUs do 4096 blocks (1 per CU slot at a time: 1024 threads) land on?
    for (int which = 0; which < 2; ++which) {
        hipStream_t s = which ? sm : sn;
        if (!s) continue;
        hipMemset(d, 0, NB * 8);
        hipLaunchKernelGGL(k_where, dim3(NB), dim3(1024), 0, s, d, 20000LL);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
        printf("%s stream: %d blocks ran on %d distinct CUs\n", which ? "masked" : "plain ", NB, distinct_cus(h, NB));
    }
    // latency of a small kernel on another stream while a long kernel saturates the (masked / plain) stream
    unsigned* d2; hipMalloc(&d2, 64 * 4);
    for (int which = 0; which < 3; ++which) {
        hipStream_t bulk = which == 1 ? sm : sn;
        if (!bulk) continue;
        hipStream_t small = which == 2 ? sp : sn == bulk ? sp : sn;
        if (which == 0) small = sp;     // plain bulk, high-priority small
        float worst = 0, sum = 0;
        const int reps = 20;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // 16384 blocks x 256 threads, ~60 us each: 8 blocks per CU at a time
        hipLaunchKernelGGL(k_where, dim3(32768), dim3(256), 0, bulk, d, 120000LL);
        std::this_thread::sleep_for(std::chrono::microseconds(300));
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(e0, small);
            hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, small, d2);
            hipEventRecord(e1, small);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            sum += ms; if (ms > worst) worst = ms;
        }
        hipDeviceSynchronize();
        printf("small kernel beside a saturating kernel on a %s stream: mean %.1f us, worst %.1f us\n",
               which == 1 ? "CU-MASKED" : "plain", sum / reps * 1e3, worst * 1e3);
    }
    return 0;
}
