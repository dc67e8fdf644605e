// sgemm_wide.hip — K4's phased far update with LDS-DMA operands and XCD-dealt tiles (round 6).
//
// The product:  C -= A^T B  phase by phase (phase = 128 k), A = a column group's error columns [Kd x M] k-major, B = the same
// rows of the inverse factor [Kd x N] k-major, C = the weight columns still to come (gptq.py:240-244 applied lazily: per element
// the same subtractions in the same order as one update per 128-column block). Same arithmetic as k_sgemm (sgemm.h): ONE
// accumulator per element and phase, products added in ascending k from +0 by v_mfma_f32_32x32x2_f32, then one rounding C - acc.
//
// Why another kernel: tools/probes/mfma_f32_peak.hip measures 0.99 of the 157.3-TFLOP/s fp32 MFMA peak with nothing else in the
// loop (one to four waves per SIMD, operands from LDS or not); k_sgemm reaches 0.58-0.72 on these shapes whatever its occupancy
// or K-step (profiles/r04_chain_pmc.txt, r05_sgemm_far_experiment.txt): its operands go through registers (83 VALU instructions
// per K-step: addresses, transposing writes, the phase end) and its tiles are dealt row by row over the XCDs. Here:
//   - operands by LDS-DMA (buffer_load_dwordx4 .. lds) into a four-slot ring of 16-k stages, two stages ahead: no staging
//     registers, no address arithmetic, no ds_write; MB + 2 DMA instructions per wave and stage. The loop is one MFMA and one
//     ds_read_b32 per slot;
//   - the C tile stays in registers; a phase's accumulators start from the inline constant +0 and are subtracted from C block by
//     block under the next phase's first MFMAs;
//   - a 1-D grid dealt so that the workgroups an XCD runs at a time are one 1024 x 1024 block of tiles (12-16 panels in that XCD's
//     L2 instead of one pair per tile), blocks of one B panel on neighbouring XCDs;
//   - the k-rows an MFMA operand read touches (lanes 0-31: k, lanes 32-63: k + 1) lie 1152 B apart in LDS: the other half of the banks.
// Two forms (template MB): 256 x 128 tiles with one workgroup per CU (128 x 64 per wave: accumulators + C = 256 of a wave's 512
// registers), and 128 x 128 tiles with two workgroups per CU, whose partner covers a workgroup's first DMA round trip, C loads,
// last subtraction and stores. Measured (profiles/r06_sgemm_wide_ab.txt): 0.74-0.87 of peak for the second form, 0.65-0.81 for the
// first, 0.59-0.72 for k_sgemm; inside the column loop -17 % / -11 % of k_sgemm's time. The second form is the default.
#include <type_traits>

#include "mfma_common.h"
#include "sgemm.h"

namespace llmc {
namespace {

constexpr int W_BN = 128, W_K = 16, W_SLOTS = 4;
constexpr int W_ROW = 1152;                    // LDS pitch of a 1-KiB piece: 1024 + 128, so that the piece holding k + 1 starts in the other half of the banks
constexpr int W_BB = (W_K / 2) * W_ROW;        // a 128-wide operand's part of a stage: 8 pieces = rows (4q + e, 4q + e + 2), q = 0..3, e = 0..1
constexpr int W_PHASE = 128 / W_K;             // stages per phase
// MB = 32-row blocks per wave along M: 4 -> 256 x 128 workgroup tile, 407 registers, one workgroup per CU, an XCD's 32 tiles = 4 x 8;
//                                      2 -> 128 x 128, two workgroups per CU (one covers the other's first and last microseconds), 64 tiles = 8 x 8
template <int MB> struct Wide {
    static constexpr int BM = 64 * MB;
    static constexpr int AB = MB == 4 ? W_K * W_ROW : W_BB;      // 256 wide: one k-row per piece; 128 wide: as B
    static constexpr int SLOT = AB + W_BB;
    static constexpr int LDS = W_SLOTS * SLOT;                   // 110592 / 73728
    static constexpr int PER_XCD = MB == 4 ? 32 : 64;            // tiles an XCD runs at a time = one block of 2^sm x 2^sn tiles (host's choice)
    static constexpr int D = MB + 2;                             // DMA instructions per wave and stage
    static constexpr int NBLK = 2 * MB;                          // accumulator blocks per wave
};

struct WideArgs {
    const float* A;
    const float* B;
    float* C;
    int64_t ldc;
    uint32_t rowA, rowB, rowC;          // bytes between k-rows of A, of B, between rows of C
    uint32_t bytesA, bytesB, bytesC;    // buffer extents from a tile's first element
    int nst;                            // Kd / 16
    int tm, tn, sbm, nsb;               // tiles along M, N; tile blocks along M; tile blocks
    int sm_log, sn_log;                 // a tile block = 2^sm_log x 2^sn_log tiles
};

template <int I, int N, typename F> __device__ __forceinline__ void wfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wfor<I + 1, N>(f);
    }
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int MB>
__global__ __launch_bounds__(256, MB == 4 ? 1 : 2) void k_sgemm_wide(const WideArgs a) {
    using W = Wide<MB>;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    // workgroup w runs on XCD w % 8 (round-robin dispatch); the PER_XCD consecutive workgroups of an XCD are one block of tiles
    const int w = blockIdx.x;
    const int g = (w / (8 * W::PER_XCD)) * 8 + (w & 7);
    if (g >= a.nsb) return;
    const int within = (w >> 3) % W::PER_XCD;
    const int ti = ((g % a.sbm) << a.sm_log) + (within >> a.sn_log), tj = ((g / a.sbm) << a.sn_log) + (within & ((1 << a.sn_log) - 1));
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
    const auto dA = mk(a.A + (int64_t)ti * W::BM, a.bytesA);
    const auto dB = mk(a.B + (int64_t)tj * W_BN, a.bytesB);
    const auto dC = mk(a.C + (int64_t)ti * W::BM * a.ldc + (int64_t)tj * W_BN, a.bytesC);
    LDS_AS char* lds = (LDS_AS char*)smem_w;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int nst = a.nst;

    // ---- LDS-DMA. A 128-wide operand: piece (q, e) = k-rows 4q + e (lanes 0-31) and 4q + e + 2 (lanes 32-63), 512 B each, at
    // (2q + e) * W_ROW; wave wv brings k-quad q = wv. The 256-wide A: one k-row (1 KiB) per piece at k * W_ROW, wave wv brings rows 4 wv ..
    const uint32_t voA = MB == 4 ? (uint32_t)lane * 16u : (uint32_t)(lane >> 5) * 2u * a.rowA + (uint32_t)(lane & 31) * 16u;
    const uint32_t voB = (uint32_t)(lane >> 5) * 2u * a.rowB + (uint32_t)(lane & 31) * 16u;
    auto dma = [&](const decltype(dA)& d, uint32_t vo, uint32_t so, uint32_t dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :: "v"(vo), "s"(d), "s"(dst), "s"(so) : "memory");
    };
    auto issue = [&](int j) {
        const uint32_t slot = lds0 + (uint32_t)(j & (W_SLOTS - 1)) * W::SLOT;
        const uint32_t k = (uint32_t)(j * W_K + 4 * wv);
        if constexpr (MB == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma(dA, voA, (k + i) * a.rowA, slot + (uint32_t)(4 * wv + i) * W_ROW);
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) dma(dA, voA, (k + e) * a.rowA, slot + (uint32_t)(2 * wv + e) * W_ROW);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) dma(dB, voB, (k + e) * a.rowB, slot + W::AB + (uint32_t)(2 * wv + e) * W_ROW);
    };

    // ---- operand reads: pair kp of a stage = k-rows 2 kp (lanes 0-31) and 2 kp + 1 (lanes 32-63); two base registers per operand
    // (slots 0-1 / 2-3: the ds_read offset field has 16 bits)
    LDS_AS char* pA[2];
    LDS_AS char* pB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        pA[h] = lds + h * 2 * W::SLOT + (lane >> 5) * W_ROW + (wm * 32 * MB + (lane & 31)) * 4;
        pB[h] = lds + h * 2 * W::SLOT + W::AB + (lane >> 5) * W_ROW + (wn * 64 + (lane & 31)) * 4;
    }
    float fa[2][MB], fb[2][2];
    auto rd = [&](auto cc, auto slc, auto kpc, auto ic) {
        constexpr int c = decltype(cc)::value, SL = decltype(slc)::value, kp = decltype(kpc)::value, i = decltype(ic)::value;
        constexpr int narrow = (SL & 1) * W::SLOT + (kp >> 1) * 2 * W_ROW + (kp & 1) * 512;
        if constexpr (i < MB) {
            if constexpr (MB == 4) fa[c][i] = *(LDS_AS const float*)(pA[SL >> 1] + (SL & 1) * W::SLOT + kp * 2 * W_ROW + i * 128);
            else fa[c][i] = *(LDS_AS const float*)(pA[SL >> 1] + narrow + i * 128);
        } else {
            fb[c][i - MB] = *(LDS_AS const float*)(pB[SL >> 1] + narrow + (i - MB) * 128);
        }
    };

    // ---- the C tile of this wave: block (m, n) element r of lane l = row wm*32*MB + m*32 + (r & 3) + 8 (r >> 2) + 4 (l >> 5),
    // column wn*64 + n*32 + (l & 31)
    const uint32_t voC = (uint32_t)(wm * 32 * MB + 4 * (lane >> 5)) * a.rowC + (uint32_t)(wn * 64 + (lane & 31)) * 4u;
    auto soC = [&](uint32_t rowC, int m, int n, int r) { return (uint32_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * rowC + (uint32_t)n * 128u; };
    f32x16 cv[MB][2], acc[MB][2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // prologue: three stages requested (nst >= 8), the first one published
    issue(0);
    issue(1);
    issue(2);
    vm_wait<2 * W::D>();
    __builtin_amdgcn_s_barrier();
    wfor<0, MB + 2>([&](auto ic) { rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ic); });
    __builtin_amdgcn_sched_barrier(0);

    // One phase = 8 stages of 8 pairs of 2 MB MFMAs. Behind MFMA i of pair p: i < MB + 2 -> operand i of the next pair (pair 7: the
    // next stage's first pair; the barrier in pair 5 has published that stage). Pair 5, last MFMA: this wave's pieces of stage j + 1
    // have landed, barrier (every wave is past its last read of stage j - 1). Pair 6, last MFMA: stage j + 3 requested into the slot
    // of stage j - 1. First phase, pair 7 of stages s < 2 MB: the C values of block s requested, i.e. issue order
    // D0 D1 D2 | D3 C0 | D4 C1 | ..: when stage j waits for D(j+1) the younger requests are C(j-2) D(j+2) C(j-1).
    constexpr int LASTI = 2 * MB - 1, CPER = 16 / (2 * MB);
    auto phase = [&](auto firstc, int j0) {
        constexpr bool FIRST = decltype(firstc)::value;
        wfor<0, W_PHASE>([&](auto sc) {
            constexpr int s = decltype(sc)::value, SL = s & (W_SLOTS - 1), SN = (s + 1) & (W_SLOTS - 1);
            const int j = j0 + s;
            wfor<0, 8>([&](auto pc) {
                constexpr int p = decltype(pc)::value, c = p & 1;
                wfor<0, 2 * MB>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, m = i >> 1, n = (m & 1) ? 1 - (i & 1) : (i & 1);
                    if constexpr (s == 0 && p == 0) {
                        // a phase's first product starts from +0 (inline constant: no zeroing); the previous phase's block is
                        // subtracted from C just before its accumulator is overwritten, under the MFMA issued before it
                        if constexpr (!FIRST) cv[m][n] = cv[m][n] - acc[m][n];
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][m], fb[c][n], zero, 0, 0, 0);
                    } else {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][m], fb[c][n], acc[m][n], 0, 0, 0);
                    }
                    if constexpr (i < MB + 2) {
                        if constexpr (p < 7) rd(std::integral_constant<int, c ^ 1>{}, std::integral_constant<int, SL>{}, std::integral_constant<int, (p + 1) & 7>{}, ic);
                        else rd(std::integral_constant<int, c ^ 1>{}, std::integral_constant<int, SN>{}, std::integral_constant<int, 0>{}, ic);
                    }
                    if constexpr (p == 5 && i == LASTI) {
                        if (j + 2 < nst) {
                            constexpr int younger_c = FIRST ? 16 * ((s >= 2 && s - 2 < W::NBLK) + (s >= 1 && s - 1 < W::NBLK)) : 0;
                            vm_wait<W::D + younger_c>();
                        } else {
                            vm_wait<0>();
                        }
                        __builtin_amdgcn_s_barrier();
                    }
                    if constexpr (p == 6 && i == LASTI) {
                        if (j + 3 < nst) issue(j + 3);
                    }
                    if constexpr (FIRST && p == 7 && s < W::NBLK) {
#pragma unroll
                        for (int e = 0; e < CPER; ++e) {
                            constexpr int bm = s >> 1, bn = s & 1;
                            cv[bm][bn][CPER * i + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dC, voC, soC(a.rowC, bm, bn, CPER * i + e), 0));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        if constexpr (FIRST) vm_wait<0>();      // the C values are here (and every request older than them)
        __builtin_amdgcn_sched_barrier(0);
    };
    phase(std::true_type{}, 0);
    for (int j0 = W_PHASE; j0 < nst; j0 += W_PHASE) phase(std::false_type{}, j0);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) cv[m][n] = cv[m][n] - acc[m][n];

    uint32_t rowC2 = a.rowC;      // opaque copy: the row offsets are recomputed here, not kept in SGPRs from the first phase on
    asm volatile("" : "+s"(rowC2));
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = cv[m][n][r];      // (a bit_cast applied to the vector element itself reads element 0: hipcc 7.2)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), dC, voC, soC(rowC2, m, n, r), 0);
            }
}

}  // namespace

// form: 0 = none, 4 = 256 x 128 tiles (one workgroup per CU), 2 = 128 x 128 tiles (two per CU)
static int wide_form(const SgemmArgs& a, bool TA, bool TB) {
    if (!TA || TB || a.batch != 1 || a.epilogue != SG_SUB || a.phase_len != 128) return 0;
    if (a.a_upper || a.a_lower || a.b_upper || a.c_upper_only) return 0;
    if (a.M <= 0 || a.N <= 0 || a.M % 128 || a.N % W_BN || a.Kd % 128 || a.Kd < 128) return 0;
    if ((a.lda % 4) || (a.ldb % 4) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 3)) return 0;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)a.Kd * a.lda * 4 >= lim || (int64_t)a.Kd * a.ldb * 4 >= lim || (int64_t)256 * a.ldc * 4 >= lim) return 0;
    if ((const void*)a.C == (const void*)a.A || (const void*)a.C == (const void*)a.B) return 0;
    const int v = opt(OPT_SGEMM_NO_WIDE);      // 0: the default form, 1: never, 2 / 4: that form where the shape allows it
    if (v == 1) return 0;
    if (v == 4) return a.M % 256 ? 2 : 4;
    return 2;      // measured (profiles/r06_sgemm_wide_ab.txt): two 128 x 128 workgroups per CU beat one 256 x 128 on every shape of the column loop
}
bool sgemm_wide_eligible(const SgemmArgs& a, bool TA, bool TB) { return wide_form(a, TA, TB) != 0; }

template <int MB> static int wide_launch(const SgemmArgs& a, hipStream_t st) {
    using W = Wide<MB>;
    WideArgs w{};
    w.A = a.A; w.B = a.B; w.C = a.C; w.ldc = a.ldc;
    w.rowA = (uint32_t)(a.lda * 4); w.rowB = (uint32_t)(a.ldb * 4); w.rowC = (uint32_t)(a.ldc * 4);
    w.bytesA = (uint32_t)(((int64_t)(a.Kd - 1) * a.lda + W::BM) * 4);
    w.bytesB = (uint32_t)(((int64_t)(a.Kd - 1) * a.ldb + W_BN) * 4);
    w.bytesC = (uint32_t)(((int64_t)(W::BM - 1) * a.ldc + W_BN) * 4);
    w.nst = a.Kd / W_K;
    w.tm = a.M / W::BM; w.tn = a.N / W_BN;
    // The shape of an XCD's tile block: the one whose busiest XCD has the fewest tiles (block g goes to XCD g % 8; a ragged last
    // block column or a block count that is not a multiple of 8 leaves XCDs idle in the last round — the column loop's far updates
    // shrink by four tile columns per launch, most of them are a few rounds long), the squarest among equals (fewest panels in L2).
    int best_cost = 1 << 30, best_sm = 0;
    constexpr int LOGT = MB == 4 ? 5 : 6;
    for (int sm = 1; sm < LOGT; ++sm) {
        const int SM = 1 << sm, SN = 1 << (LOGT - sm);
        const int sbm = (w.tm + SM - 1) / SM, sbn = (w.tn + SN - 1) / SN;
        int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < sbm * sbn; ++g) {
            const int bi = g % sbm, bj = g / sbm;
            load[g & 7] += (w.tm - bi * SM < SM ? w.tm - bi * SM : SM) * (w.tn - bj * SN < SN ? w.tn - bj * SN : SN);
        }
        int mx = 0;
        for (int x = 0; x < 8; ++x) mx = load[x] > mx ? load[x] : mx;
        const int cost = mx * 64 + (SM + SN);
        if (cost < best_cost) { best_cost = cost; best_sm = sm; }
    }
    w.sm_log = best_sm; w.sn_log = LOGT - best_sm;
    w.sbm = (w.tm + (1 << w.sm_log) - 1) >> w.sm_log;
    w.nsb = w.sbm * ((w.tn + (1 << w.sn_log) - 1) >> w.sn_log);
    const int rounds = (w.nsb + 7) / 8;
    if (int rc = ensure_dynamic_lds((const void*)k_sgemm_wide<MB>, W::LDS)) return rc;
    hipLaunchKernelGGL(k_sgemm_wide<MB>, dim3(rounds * 8 * W::PER_XCD), dim3(256), W::LDS, st, w);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
int sgemm_wide_launch(const SgemmArgs& a, hipStream_t st) {
    return wide_form(a, true, false) == 4 ? wide_launch<4>(a, st) : wide_launch<2>(a, st);
}

}  // namespace llmc
