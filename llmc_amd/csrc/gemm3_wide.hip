// gemm3_wide.hip — K3's far update C -= A^T B on pre-split bf16 planes, two 128 x 128 workgroups per CU (round 6).
//
// The product (cholesky.hip, the symmetric update of the trailing matrix behind an outer block, gptq.py:172-174): A = B = the block's
// panel P [Kd x n] k-major, split ONCE into its hi | mid | lo bf16 planes (k_split3_planes); C -= P[:, rows]^T P[:, cols] with six bf16
// products per fp32 product, in k_gemm3's order per accumulator (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; k ascending): the
// same bits as k_gemm3 / k_gemm3s (tests/test_gptq_gpu.py compares them).
//
// Why another kernel: k_gemm3s (one 512-thread workgroup per CU: four MFMA waves + four producer waves, 256 x 128 tiles) spends ~2100
// cycles on a 1536-cycle K-step — the step's first fragments cannot be read before the barrier that publishes them, and with two
// waves per SIMD there are no registers to read them a step ahead — and ~11k of a tile's 82k cycles before its first and after its
// last MFMA (profiles/r06_gemm3s_dma.txt). The structure that took K4's far update from 0.62 to 0.74-0.87 of its pipe's peak
// (sgemm_wide.hip) applied to the planes form:
//   - 128 x 128 tiles, 256 threads (64 x 64 per wave: four accumulator blocks), TWO workgroups per CU: one covers the other's first
//     DMA round trip and its C update (load, subtract, store at the tile's end: no C registers in the loop);
//   - no producer waves: the MFMA waves request their stage's planes by LDS-DMA (six 1-KiB pieces per wave and 16-k stage) into a
//     THREE-slot ring (24 KiB per stage: 72 KiB per workgroup);
//   - ALL twelve fragments of stage j + 1 are read under the first twelve of stage j's 24 MFMAs (fragment registers double-buffered:
//     96 + 64 accumulator registers of the 256 a wave may have); so a slot is free again at the barrier that
//     opens its stage, stage j + 3 is requested there and has two stages (~3000 cycles with both workgroups on the pipe) to land;
//   - a 1-D grid dealt so that the 64 workgroups an XCD runs at a time are one block of tiles (of the shape that leaves the
//     busiest XCD the fewest tiles that do work — the update is the upper triangle), as in sgemm_wide.hip.
// LDS image of a plane's stage: 16 k-rows of 256 B, 64-B units XOR-swizzled by (k & 3) exactly as k_gemm3s's B planes (the swizzle
// is applied to the DMA's per-lane source column; ds_read_b64_tr_b16 fragment reads undo it).
#include <type_traits>

#include "mfma_common.h"
#include "sgemm.h"

namespace llmc {
namespace {

constexpr int G_B = 128;                      // tile edge
constexpr int G_K = 16;                       // stage depth = one MFMA
constexpr int G_ROW = G_B * 2;                // bytes per k-row of a plane
constexpr int G_PLANE = G_K * G_ROW;          // 4 KiB
constexpr int G_OPND = 3 * G_PLANE;           // 12 KiB
constexpr int G_SLOT = 2 * G_OPND;            // 24 KiB
constexpr int G_SLOTS = 3;
constexpr int G_LDS = G_SLOTS * G_SLOT;       // 72 KiB
constexpr int G_PER_XCD = 64;                 // tiles an XCD runs at a time (32 CUs x 2)

struct G3wArgs {
    const uint16_t* PA;
    const uint16_t* PB;
    float* C;
    int64_t ldc;
    uint32_t row2, ps2, rowC;         // bytes between k-rows of a plane, between planes, between rows of C
    uint32_t bytesP, bytesC;
    int nst;                          // Kd / 16 (even)
    int tm, tn, sbm, nsb, sm_log, sn_log;
    int upper;                        // c_upper_only
};

template <int I, int N, typename F> __device__ __forceinline__ void gfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        gfor<I + 1, N>(f);
    }
}
template <int N> __device__ __forceinline__ void g_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ s16x8 g_frag(LDS_AS char* p, int imm) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm + 4 * G_ROW));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void k_gemm3w(const G3wArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    const int w = blockIdx.x;
    const int g = (w / (8 * G_PER_XCD)) * 8 + (w & 7);
    if (g >= a.nsb) return;
    const int within = (w >> 3) % G_PER_XCD;
    const int ti = ((g % a.sbm) << a.sm_log) + (within >> a.sn_log), tj = ((g / a.sbm) << a.sn_log) + (within & ((1 << a.sn_log) - 1));
    if (ti >= a.tm || tj >= a.tn) return;
    if (a.upper && tj < ti) return;               // tile strictly below the diagonal

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
#if defined(G3W_DBG) && G3W_DBG == 1      // lab (tools/probes/g3w_lab.sh): every tile reads the panels of tile (0, 0) — wrong results, perfect L2 reuse
    const auto dA = mk(a.PA, a.bytesP);
    const auto dB = mk(a.PB, a.bytesP);
#else
    const auto dA = mk(a.PA + (int64_t)ti * G_B, a.bytesP);
    const auto dB = mk(a.PB + (int64_t)tj * G_B, a.bytesP);
#endif
    const auto dC = mk(a.C + (int64_t)ti * G_B * a.ldc + (int64_t)tj * G_B, a.bytesC);
    LDS_AS char* lds = (LDS_AS char*)smem_g;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int nst = a.nst;

    // ---- LDS-DMA: a piece = four k-rows (256 B each, 16 lanes a row) of one plane; a stage has 2 operands x 3 planes x 4 pieces = 24,
    // wave wv brings k-quad q = wv of all six (operand, plane) pairs
    uint32_t vo;
    {
        const int r = lane >> 4, c = lane & 15;
        const int cs = (((c >> 2) ^ r) << 2) | (c & 3);          // the row's 64-B units XOR (k & 3), k & 3 = r
        vo = (uint32_t)r * a.row2 + (uint32_t)cs * 16u;
    }
    auto dma = [&](const decltype(dA)& d, uint32_t so, uint32_t dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :: "v"(vo), "s"(d), "s"(dst), "s"(so) : "memory");
    };
    auto issue = [&](int j, uint32_t slot_off) {
        const uint32_t k0 = (uint32_t)j * G_K;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int op = t / 3, pl = t % 3;
            const uint32_t so = (uint32_t)pl * a.ps2 + (k0 + 4u * (uint32_t)wv) * a.row2;
            const uint32_t dst = lds0 + slot_off + (uint32_t)op * G_OPND + (uint32_t)pl * G_PLANE + (uint32_t)wv * 1024u;
            if (op == 0) dma(dA, so, dst); else dma(dB, so, dst);
        }
    };

    // ---- fragment reads (k_gemm3s's B-plane addressing): block b of this wave's operand half
    const int p = lane & 15;
    const int trow = 8 * (lane >> 5) + (p >> 2);
    const int sub = 32 * ((lane >> 4) & 1) + 8 * (p & 3);
    int offA[2], offB[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        offA[b] = trow * G_ROW + (((2 * wm + b) ^ (p >> 2)) << 6) + sub;
        offB[b] = G_OPND + trow * G_ROW + (((2 * wn + b) ^ (p >> 2)) << 6) + sub;
    }
    s16x8 fa[2][2][3], fb[2][2][3];      // [set][block][plane]
    // fragment f (in the order the products need them: hi of B and lo of A first) of the stage whose slot starts at `base`
    auto rd = [&](auto setc, auto fc, LDS_AS char* base) {
        constexpr int S = decltype(setc)::value, f = decltype(fc)::value;
        constexpr int grp = f >> 2, e = f & 3, b = e & 1;                  // groups: {B hi, A lo} {A hi, B lo} {A mid, B mid}
        constexpr bool isA = grp == 0 ? e >= 2 : e < 2;
        constexpr int pl = grp == 0 ? (isA ? 2 : 0) : grp == 1 ? (isA ? 0 : 2) : 1;
        if constexpr (isA) fa[S][b][pl] = g_frag(base + offA[b], pl * G_PLANE);
        else fb[S][b][pl] = g_frag(base + offB[b], pl * G_PLANE);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // prologue: three stages requested, the first one published and read
    issue(0, 0);
    issue(1, G_SLOT);
    issue(2, 2 * G_SLOT);
    g_vm_wait<12>();
    __builtin_amdgcn_s_barrier();
    gfor<0, 12>([&](auto fc) { rd(std::integral_constant<int, 0>{}, fc, lds); });
    __builtin_amdgcn_sched_barrier(0);

    // Stage j (fragment set S = j & 1), its planes in slot j % 3: this wave's pieces of stage j + 1 have landed (stage j + 2's six may
    // fly); barrier: every wave has read stage j (under stage j - 1's MFMAs) and stage j + 1 is complete; stage j + 3 requested into
    // stage j's slot; then 24 MFMAs = six products x four accumulators in rotation, a fragment of stage j + 1 read behind each of the
    // first twelve (all of them back long before the next barrier: the slot they came from is refilled behind it).
    // The old C values of block cb are requested under the MFMAs of stage cb (cb = 0..3: two loads behind each of MFMAs 12..19), so
    // that the tile ends with a subtraction and stores only (lab builds, tools/probes/g3w_lab.sh: a load-subtract-store epilogue was
    // 17 % of the kernel — one workgroup alone does not fill the pipe while its partner waits for C). Issue order D0 D1 D2 | D3 C0 |
    // D4 C1 | D5 C2 | D6 C3 | D7 ..: the requests younger than D(j+1) when stage j waits for it are C(j-2) D(j+2) C(j-1).
    const uint32_t voC = (uint32_t)(wm * 64 + 4 * (lane >> 5)) * a.rowC + (uint32_t)(wn * 64 + (lane & 31)) * 4u;
    auto soC = [&](uint32_t rowC, int m, int n, int r) { return (uint32_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * rowC + (uint32_t)n * 128u; };
    f32x16 oldc[2][2];
    uint32_t cur = 0;                                   // byte offset of stage j's slot
    auto stage = [&](auto setc, auto cbc, auto youngc, int j) {
        constexpr int S = decltype(setc)::value, CB = decltype(cbc)::value, YOUNG = decltype(youngc)::value;
        if (j + 2 < nst) g_vm_wait<YOUNG>(); else g_vm_wait<0>();
        __builtin_amdgcn_s_barrier();
#if !(defined(G3W_DBG) && G3W_DBG == 2)    // lab: no operand traffic after the prologue
        if (j + 3 < nst) issue(j + 3, cur);
#endif
        const uint32_t nxt = cur + G_SLOT == G_LDS ? 0u : cur + G_SLOT;
        LDS_AS char* nb = lds + nxt;
        __builtin_amdgcn_sched_barrier(0);
        gfor<0, 24>([&](auto ic) {
            constexpr int i = decltype(ic)::value, q = i >> 2, m = (i >> 1) & 1, n = i & 1;
            constexpr int TA = q == 0 ? 2 : q == 1 ? 0 : q == 2 ? 1 : q == 3 ? 1 : 0;      // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
            constexpr int TB = q == 0 ? 0 : q == 1 ? 2 : q == 2 ? 1 : q == 3 ? 0 : q == 4 ? 1 : 0;
            acc[m][n] = Mfma<LLMC_BF16>::run(fa[S][m][TA], fb[S][n][TB], acc[m][n]);
            if constexpr (i < 12) rd(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, i>{}, nb);
#if !(defined(G3W_DBG) && G3W_DBG == 3)    // lab: no C traffic
            if constexpr (CB >= 0 && i >= 12 && i < 20) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    constexpr int r0 = 2 * (i - 12);
                    oldc[CB >> 1][CB & 1][r0 + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dC, voC, soC(a.rowC, CB >> 1, CB & 1, r0 + e), 0));
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        cur = nxt;
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    constexpr std::integral_constant<int, -1> NOC{};
    stage(S0, std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, 0);          // nst >= 8
    stage(S1, std::integral_constant<int, 1>{}, std::integral_constant<int, 22>{}, 1);
    stage(S0, std::integral_constant<int, 2>{}, std::integral_constant<int, 38>{}, 2);
    stage(S1, std::integral_constant<int, 3>{}, std::integral_constant<int, 38>{}, 3);
    stage(S0, NOC, std::integral_constant<int, 38>{}, 4);
    stage(S1, NOC, std::integral_constant<int, 22>{}, 5);
    for (int j = 6; j < nst; j += 2) {
        stage(S0, NOC, std::integral_constant<int, 6>{}, j);
        stage(S1, NOC, std::integral_constant<int, 6>{}, j + 1);
    }

    // ---- C = old - acc, block by block; of a diagonal tile only the blocks that reach the diagonal (k_gemm3s's row limit)
    uint32_t rowC2 = a.rowC;      // opaque copy: the row offsets are recomputed here, not kept in SGPRs from the first stages on
    asm volatile("" : "+s"(rowC2));
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (a.upper && ti == tj && 2 * wn + n < 2 * wm + m) continue;
#if defined(G3W_DBG) && G3W_DBG == 3
            if (acc[m][n][0] != 12345.678f) continue;
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = oldc[m][n][r] - acc[m][n][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), dC, voC, soC(rowC2, m, n, r), 0);
            }
        }
}

}  // namespace

bool gemm3w_eligible(const SgemmArgs& a) {
    if (opt(OPT_GEMM3_NO_WIDE) || opt(OPT_GEMM3_NOSPEC)) return false;
    if (!a.planesA || !a.planesB || a.batch != 1 || a.epilogue != SG_SUB || a.phase_len != 0) return false;
    if (a.a_upper || a.a_lower || a.b_upper) return false;
    if (a.M <= 0 || a.N <= 0 || a.M % G_B || a.N % G_B || a.Kd % (2 * G_K) || a.Kd < 8 * G_K) return false;
    if (a.ldp % 8 || a.plane_stride % 8 || (((uintptr_t)a.planesA | (uintptr_t)a.planesB) & 15) || ((uintptr_t)a.C & 3)) return false;
    if ((2 * a.plane_stride + ((int64_t)a.Kd + G_K) * a.ldp) * 2 >= (int64_t)0x7fffff00) return false;
    if ((int64_t)G_B * a.ldc * 4 >= (int64_t)0x7fffff00) return false;
    const int64_t tm = a.M / G_B, tn = a.N / G_B;
    const int64_t tiles = a.c_upper_only ? tm * tn - tm * (tm - 1) / 2 : tm * tn;
    const int mt = opt(OPT_GEMM3S_MIN_TILES);
    return tiles >= (mt > 0 ? mt : 1024);      // measured (profiles/r06_gemm3w_ab.txt): n = 3584 (406 tiles) 68 vs 54 us for k_gemm3s, n = 8192 (2080) 236-260 vs 268
}

int gemm3w_launch(const SgemmArgs& a, hipStream_t st) {
    G3wArgs w{};
    w.PA = (const uint16_t*)a.planesA; w.PB = (const uint16_t*)a.planesB; w.C = a.C; w.ldc = a.ldc;
    w.row2 = (uint32_t)(a.ldp * 2); w.ps2 = (uint32_t)(a.plane_stride * 2); w.rowC = (uint32_t)(a.ldc * 4);
    w.bytesP = (uint32_t)((2 * a.plane_stride + (int64_t)(a.Kd - 1) * a.ldp + G_B) * 2);
    w.bytesC = (uint32_t)(((int64_t)(G_B - 1) * a.ldc + G_B) * 4);
    w.nst = a.Kd / G_K;
    w.tm = a.M / G_B; w.tn = a.N / G_B;
    w.upper = a.c_upper_only ? 1 : 0;
    // the tile-block shape whose busiest XCD has the fewest WORKING tiles (block g -> XCD g % 8), the squarest among equals
    int best_cost = 1 << 30, best_sm = 3;
    for (int sm = 1; sm < 6; ++sm) {
        const int SM = 1 << sm, SN = 1 << (6 - sm);
        const int sbm = (w.tm + SM - 1) / SM, sbn = (w.tn + SN - 1) / SN;
        int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < sbm * sbn; ++g) {
            const int i0 = (g % sbm) * SM, j0 = (g / sbm) * SN;
            int n = 0;
            for (int i = i0; i < i0 + SM && i < w.tm; ++i) {
                const int jlo = w.upper && i > j0 ? i : j0, jhi = j0 + SN < w.tn ? j0 + SN : w.tn;
                if (jhi > jlo) n += jhi - jlo;
            }
            load[g & 7] += n;
        }
        int mx = 0;
        for (int x = 0; x < 8; ++x) mx = load[x] > mx ? load[x] : mx;
        const int cost = mx * 64 + (SM + SN);
        if (cost < best_cost) { best_cost = cost; best_sm = sm; }
    }
    w.sm_log = best_sm; w.sn_log = 6 - best_sm;
    w.sbm = (w.tm + (1 << w.sm_log) - 1) >> w.sm_log;
    w.nsb = w.sbm * ((w.tn + (1 << w.sn_log) - 1) >> w.sn_log);
    const int rounds = (w.nsb + 7) / 8;
    if (int rc = ensure_dynamic_lds((const void*)k_gemm3w, G_LDS)) return rc;
    hipLaunchKernelGGL(k_gemm3w, dim3(rounds * 8 * G_PER_XCD), dim3(256), G_LDS, st, w);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

}  // namespace llmc
