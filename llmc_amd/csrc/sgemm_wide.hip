// sgemm_wide.hip — K4's phased far update on a 256 x 128 workgroup tile with ONE wave per SIMD (round 6).
//
// The product:  C -= A^T B  phase by phase (phase = 128 k), A = a column group's error columns [Kd x M] k-major, B = the same
// rows of the inverse factor [Kd x N] k-major, C = the weight columns still to come (gptq.py:240-244 applied lazily: per element
// the same subtractions in the same order as one update per 128-column block). Same arithmetic as k_sgemm (sgemm.h): ONE
// accumulator per element and phase, products added in ascending k from +0 by v_mfma_f32_32x32x2_f32, then one rounding C - acc.
//
// Why another kernel: tools/probes/mfma_f32_peak.hip measures 0.99 of the 157.3-TFLOP/s fp32 MFMA peak with nothing else in the
// loop (one to four waves per SIMD, operands from LDS or not), k_sgemm reaches 0.61-0.66 on these shapes whatever its occupancy
// or K-step (profiles/r04_chain_pmc.txt, r05_sgemm_far_experiment.txt): its 128 x 128 tile asks the memory system for 1 B per
// 32 flop (4.9 TB/s of operand fetch at peak), through registers, with 83 VALU instructions per K-step. Here:
//   - 256 x 128 per workgroup (128 x 64 per wave: 8 accumulator blocks + the C tile itself = 256 registers of a 512-register wave):
//     1 B per 43 flop;
//   - operands by LDS-DMA (buffer_load_dwordx4 .. lds) into a four-slot ring of 16-k stages, two stages ahead: no staging
//     registers, no address arithmetic, no ds_write; six DMA instructions per wave and stage;
//   - a 1-D grid dealt so that the 32 workgroups an XCD runs at a time are a 4 x 8 block of tiles (8 + 4 panels for 32 tiles in that
//     XCD's L2 instead of one pair per tile) and the eight XCDs' blocks share their B panels;
//   - the k-rows an MFMA operand read touches (lanes 0-31: k, lanes 32-63: k + 1) lie 1152 B apart in LDS: the other half of the banks.
#include <type_traits>

#include "mfma_common.h"
#include "sgemm.h"

namespace llmc {
namespace {

constexpr int W_BM = 256, W_BN = 128, W_K = 16, W_SLOTS = 4;
constexpr int W_ROW = 1152;                    // bytes between the LDS images of consecutive A k-rows (1 KiB of data) / of the two B pieces of a k-quad
constexpr int W_AB = W_K * W_ROW;              // A part of a stage: 16 k-rows of 256 floats
constexpr int W_BB = (W_K / 2) * W_ROW;        // B part: 8 pieces of 1 KiB = rows (4q + e, 4q + e + 2), q = 0..3, e = 0..1
constexpr int W_SLOT = W_AB + W_BB;            // 27648
constexpr int W_LDS = W_SLOTS * W_SLOT;        // 110592
constexpr int W_SM = 4, W_SN = 8;              // an XCD's block of tiles
constexpr int W_PHASE = 128 / W_K;             // stages per phase

struct WideArgs {
    const float* A;
    const float* B;
    float* C;
    int64_t ldc;
    uint32_t rowA, rowB, rowC;          // bytes between k-rows of A, of B, between rows of C
    uint32_t bytesA, bytesB, bytesC;    // buffer extents from a tile's first element
    int nst;                            // Kd / 16
    int tm, tn, sbm, nsb;               // tiles along M, N; tile blocks along M; tile blocks
};

template <int I, int N, typename F> __device__ __forceinline__ void wfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wfor<I + 1, N>(f);
    }
}

__global__ __launch_bounds__(256) void k_sgemm_wide(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    // workgroup w runs on XCD w % 8 (round-robin dispatch); the 32 consecutive workgroups of an XCD are one 4 x 8 block of tiles
    const int w = blockIdx.x;
    const int g = (w >> 8) * 8 + (w & 7);
    if (g >= a.nsb) return;
    const int within = (w >> 3) & 31;
    const int ti = (g % a.sbm) * W_SM + (within >> 3), tj = (g / a.sbm) * W_SN + (within & 7);
    if (ti >= a.tm || tj >= a.tn) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    auto mk = [](const void* p, uint32_t bytes) {
        const uint64_t u = (uint64_t)p;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const auto dA = mk(a.A + (int64_t)ti * W_BM, a.bytesA);
    const auto dB = mk(a.B + (int64_t)tj * W_BN, a.bytesB);
    const auto dC = mk(a.C + (int64_t)ti * W_BM * a.ldc + (int64_t)tj * W_BN, a.bytesC);
    LDS_AS char* lds = (LDS_AS char*)smem_w;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int nst = a.nst;

    // ---- LDS-DMA: wave wv brings A k-rows 4 wv .. 4 wv + 3 (1 KiB each) and the two B pieces of k-quad wv of every stage
    const uint32_t voA = (uint32_t)lane * 16u;
    const uint32_t voB = (uint32_t)(lane >> 5) * 2u * a.rowB + (uint32_t)(lane & 31) * 16u;
    auto dma = [&](const decltype(dA)& d, uint32_t vo, uint32_t so, uint32_t dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :: "v"(vo), "s"(d), "s"(dst), "s"(so) : "memory");
    };
    auto issue = [&](int j) {
        const uint32_t slot = lds0 + (uint32_t)(j & (W_SLOTS - 1)) * W_SLOT;
        const uint32_t k = (uint32_t)(j * W_K + 4 * wv);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma(dA, voA, (k + i) * a.rowA, slot + (uint32_t)(4 * wv + i) * W_ROW);
#pragma unroll
        for (int e = 0; e < 2; ++e) dma(dB, voB, (k + e) * a.rowB, slot + W_AB + (uint32_t)(2 * wv + e) * W_ROW);
    };

    // ---- operand reads: pair kp of a stage = k-rows 2 kp (lanes 0-31) and 2 kp + 1 (lanes 32-63)
    LDS_AS char* pA[2];
    LDS_AS char* pB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        pA[h] = lds + h * 2 * W_SLOT + (lane >> 5) * W_ROW + (wm * 128 + (lane & 31)) * 4;
        pB[h] = lds + h * 2 * W_SLOT + W_AB + (lane >> 5) * W_ROW + (wn * 64 + (lane & 31)) * 4;
    }
    float fa[2][4], fb[2][2];
    auto rd = [&](auto cc, auto slc, auto kpc, auto ic) {
        constexpr int c = decltype(cc)::value, SL = decltype(slc)::value, kp = decltype(kpc)::value, i = decltype(ic)::value;
        if constexpr (i < 4) fa[c][i] = *(LDS_AS const float*)(pA[SL >> 1] + (SL & 1) * W_SLOT + kp * 2 * W_ROW + i * 128);
        else fb[c][i - 4] = *(LDS_AS const float*)(pB[SL >> 1] + (SL & 1) * W_SLOT + (kp >> 1) * 2 * W_ROW + (kp & 1) * 512 + (i - 4) * 128);
    };

    // ---- the C tile of this wave: block (m, n) element r of lane l = row wm*128 + m*32 + (r & 3) + 8 (r >> 2) + 4 (l >> 5),
    // column wn*64 + n*32 + (l & 31)
    const uint32_t voC = (uint32_t)(wm * 128 + 4 * (lane >> 5)) * a.rowC + (uint32_t)(wn * 64 + (lane & 31)) * 4u;
    auto soC = [&](uint32_t rowC, int m, int n, int r) { return (uint32_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * rowC + (uint32_t)n * 128u; };
    f32x16 cv[4][2], acc[4][2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // prologue: three stages requested, the first one published
    issue(0);
    if (1 < nst) issue(1);
    if (2 < nst) issue(2);
    if (2 < nst) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    wfor<0, 6>([&](auto ic) { rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ic); });
    __builtin_amdgcn_sched_barrier(0);

    // One phase = 8 stages = 64 pairs of 8 MFMAs. Behind MFMA i of pair p: i < 6 -> operand i of the next pair (pair 7: the next
    // stage's first pair; the barrier in pair 5 has published that stage). Pair 5, last MFMA: this wave's pieces of stage j + 1 have
    // landed, barrier (every wave is past its last read of stage j - 1). Pair 6: stage j + 3 requested into the slot of stage j - 1.
    // First phase, pair 7: the C values of block s requested (two per MFMA), i.e. issue order D0 D1 D2 | D3 C0 | D4 C1 | ..: when
    // stage j waits for D(j+1) the younger requests are C(j-2) D(j+2) C(j-1) = 38 (j = 0: D2 = 6; j = 1: D3 C0 = 22); later phases: 6.
    auto phase = [&](auto firstc, int j0) {
        constexpr bool FIRST = decltype(firstc)::value;
        wfor<0, W_PHASE>([&](auto sc) {
            constexpr int s = decltype(sc)::value, SL = s & (W_SLOTS - 1), SN = (s + 1) & (W_SLOTS - 1);
            const int j = j0 + s;
            wfor<0, 8>([&](auto pc) {
                constexpr int p = decltype(pc)::value, c = p & 1;
                wfor<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, m = i >> 1, n = (m & 1) ? 1 - (i & 1) : (i & 1);
                    if constexpr (s == 0 && p == 0) {
                        // a phase's first product starts from +0 (inline constant: no zeroing); the previous phase's block is
                        // subtracted from C just before its accumulator is overwritten, under the MFMA issued before it
                        if constexpr (!FIRST) cv[m][n] = cv[m][n] - acc[m][n];
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][m], fb[c][n], zero, 0, 0, 0);
                    } else {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][m], fb[c][n], acc[m][n], 0, 0, 0);
                    }
                    if constexpr (i < 6) {
                        if constexpr (p < 7) rd(std::integral_constant<int, c ^ 1>{}, std::integral_constant<int, SL>{}, std::integral_constant<int, (p + 1) & 7>{}, ic);
                        else rd(std::integral_constant<int, c ^ 1>{}, std::integral_constant<int, SN>{}, std::integral_constant<int, 0>{}, ic);
                    }
                    if constexpr (p == 5 && i == 7) {
                        if (j + 2 < nst) {
                            if constexpr (FIRST && s == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                            else if constexpr (FIRST && s == 1) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                            else if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        __builtin_amdgcn_s_barrier();
                    }
                    if constexpr (p == 6 && i == 6) {
                        if (j + 3 < nst) issue(j + 3);
                    }
                    if constexpr (FIRST && p == 7) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            constexpr int bm = s >> 1, bn = s & 1;
                            cv[bm][bn][2 * i + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dC, voC, soC(a.rowC, bm, bn, 2 * i + e), 0));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the C values are here (and every request older than them)
        __builtin_amdgcn_sched_barrier(0);
    };
    phase(std::true_type{}, 0);
    for (int j0 = W_PHASE; j0 < nst; j0 += W_PHASE) phase(std::false_type{}, j0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) cv[m][n] = cv[m][n] - acc[m][n];

    uint32_t rowC2 = a.rowC;      // opaque copy: the 128 row offsets are recomputed here, not kept in SGPRs from the first phase on
    asm volatile("" : "+s"(rowC2));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = cv[m][n][r];      // (a bit_cast applied to the vector element itself reads element 0: hipcc 7.2)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), dC, voC, soC(rowC2, m, n, r), 0);
            }
}

}  // namespace

bool sgemm_wide_eligible(const SgemmArgs& a, bool TA, bool TB) {
    if (!TA || TB || a.batch != 1 || a.epilogue != SG_SUB || a.phase_len != 128) return false;
    if (a.a_upper || a.a_lower || a.b_upper || a.c_upper_only) return false;
    if (a.M <= 0 || a.N <= 0 || a.M % W_BM || a.N % W_BN || a.Kd % 128 || a.Kd < 128) return false;
    if ((a.lda % 4) || (a.ldb % 4) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 3)) return false;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)a.Kd * a.lda * 4 >= lim || (int64_t)a.Kd * a.ldb * 4 >= lim || (int64_t)W_BM * a.ldc * 4 >= lim) return false;
    if ((const void*)a.C == (const void*)a.A || (const void*)a.C == (const void*)a.B) return false;
    return !opt(OPT_SGEMM_NO_WIDE);
}

int sgemm_wide_launch(const SgemmArgs& a, hipStream_t st) {
    WideArgs w{};
    w.A = a.A; w.B = a.B; w.C = a.C; w.ldc = a.ldc;
    w.rowA = (uint32_t)(a.lda * 4); w.rowB = (uint32_t)(a.ldb * 4); w.rowC = (uint32_t)(a.ldc * 4);
    w.bytesA = (uint32_t)(((int64_t)(a.Kd - 1) * a.lda + W_BM) * 4);
    w.bytesB = (uint32_t)(((int64_t)(a.Kd - 1) * a.ldb + W_BN) * 4);
    w.bytesC = (uint32_t)(((int64_t)(W_BM - 1) * a.ldc + W_BN) * 4);
    w.nst = a.Kd / W_K;
    w.tm = a.M / W_BM; w.tn = a.N / W_BN;
    w.sbm = (w.tm + W_SM - 1) / W_SM;
    w.nsb = w.sbm * ((w.tn + W_SN - 1) / W_SN);
    const int rounds = (w.nsb + 7) / 8;
    if (int rc = ensure_dynamic_lds((const void*)k_sgemm_wide, W_LDS)) return rc;
    hipLaunchKernelGGL(k_sgemm_wide, dim3(rounds * 256), dim3(256), W_LDS, st, w);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

}  // namespace llmc
