// hessian_syrk.hip — K1: H <- a*H + b*X^T X on the MFMA pipe (GPTQ.add_batch, gptq.py:254-295).
//
// X is a LIST of calibration samples, each [T_i tokens, K channels] 16-bit, channel-contiguous, anywhere in HBM (llmc's
// hooks deliver one tensor per sample, gptq.py:254-295 with calib.bs = 1); one contiguous [T, K] tensor is the list of
// length 1. Both MFMA operands are strided along the reduction (token) axis, so tiles are staged token-major in LDS
// exactly as they lie in HBM (LDS-DMA, 16 B per lane) and transposed on the way to the registers by
// ds_read_b64_tr_b16.
//
// Decomposition
//   token axis    the samples back to back, each padded to a whole number of 128-token GROUPS (one turn of the LDS
//                 ring); the padding rows are never read: they lie past the sample's buffer descriptor and come back
//                 as zeros, which add nothing.
//   output tile   256 x 256 channels (lower triangle of the tile grid only); 4 waves as 2(M) x 2(N), one per SIMD,
//                 each 128 x 128 = 4 x 4 MFMA 32x32x16 accumulators (256 accumulator registers)
//   unit          (tile, token chunk s of S); chunk boundaries are group boundaries. Units are dealt round-robin to a
//                 persistent grid of one workgroup per CU so that the 32 workgroups of an XCD sit on 32 consecutive
//                 tiles of a 4x4-superblock order (12 shared panels in the XCD's L2) at the SAME token position.
//   partials      every unit writes its 256x256 fp32 partial in MFMA-fragment order (16-B stores) to the
//                 workspace; k_syrk_fixup sums the S partials of a tile in chunk order (deterministic),
//                 applies H <- a*H + b*sum and mirrors the tile to the upper triangle.
//   sample table  a sample's rows are addressed through ITS OWN buffer descriptor whose base is shifted so that the
//                 unit-relative byte offsets the pieces already carry (stage * 32 rows + piece rows) land on the
//                 sample's rows: base' = sample base - (unit-relative offset of the sample's first group), num_records
//                 = unit-relative offset of the sample's last valid row + 1 row. The descriptor therefore changes only
//                 when the token walk crosses into the next sample, never inside one. A unit keeps the descriptor
//                 words of the (up to 64) samples it crosses in four VGPRs, one sample per lane, and pulls the next
//                 one into SGPRs with v_readlane once per group: no memory access, no branch in the stage loop.
// LDS bank layout: a 4-token x 64-B tr16 read by a 32-lane half hits 4 rows at a 512-B stride; the 64-B
// unit index is XOR-ed with (token & 3) so the four rows land on four different bank quarters. The
// swizzle is applied to the DMA's per-lane SOURCE address (LDS image stays lane-linear) and to the read.
//
// Lab builds (-DLLMC_LAB, tools/probes only; never in the shipped library): ablation instantiations (wrong results
// by design) and the LLMC_SYRK_* environment overrides.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "mfma_common.h"

namespace llmc {

static constexpr int TM = 256;       // tile edge (channels)
static constexpr int TILE_FLOATS = TM * TM;
static constexpr int GROUP_TOK = 128;            // tokens per group = one turn of the 4-slot ring of 32-token stages
static constexpr int SYRK_MAX_SAMPLES = 512;     // table entries per launch, all problems together (8 KiB of kernel arguments)
static constexpr int SYRK_MAX_CHUNKS = 32;
static constexpr int SYRK_UNIT_SAMPLES = 64;     // samples one unit may cross (one per lane; the kernel clamps at 63)
static constexpr int FOLD_GROUPS = 16;           // exact diagonal: the diagonal wave folds its accumulators into fp64 every 16 groups (2048 tokens)
static constexpr int SYRK_MAX_PROBS = 4;         // Hessians (problems) one launch may carry in its unit queue

struct TileIdx {
    int bi, bj;
    bool valid;
};

// tile order: 4x4 superblocks of the lower triangle, superblock rows top to bottom; inside an
// off-diagonal superblock row-major 4x4, inside a diagonal superblock the 10 lower-triangular tiles.
__host__ __device__ inline int tiles_padded(int nb) {
    int sb = (nb + 3) / 4;
    return 8 * sb * sb + 2 * sb;
}
__host__ __device__ inline TileIdx decode_tile(int ti, int nb) {
    int sbi = (int)((sqrtf(4.0f + 32.0f * (float)ti) - 2.0f) / 16.0f);
    while (8 * (sbi + 1) * (sbi + 1) + 2 * (sbi + 1) <= ti) ++sbi;
    while (sbi > 0 && 8 * sbi * sbi + 2 * sbi > ti) --sbi;
    int rem = ti - (8 * sbi * sbi + 2 * sbi);
    int sbj, r, c;
    if (rem < 16 * sbi) {
        sbj = rem >> 4;
        int pos = rem & 15;
        r = pos >> 2;
        c = pos & 3;
    } else {
        sbj = sbi;
        int pos = rem - 16 * sbi;
        r = pos < 1 ? 0 : (pos < 3 ? 1 : (pos < 6 ? 2 : 3));
        c = pos - r * (r + 1) / 2;
    }
    TileIdx t;
    t.bi = 4 * sbi + r;
    t.bj = 4 * sbj + c;
    t.valid = t.bi < nb;
    return t;
}

__device__ __forceinline__ s16x8 tr_frag(LDS_AS char* p, int imm0) {
    // two 4-token transposed reads -> 8 consecutive tokens of one channel per lane
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + imm0 + 4 * TM * 2));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct SyrkSample {
    uint64_t base;   // device address of the sample's first row
    uint32_t T;      // its tokens (> 0)
    uint32_t g0;     // its first group on the padded token axis
};

// One Hessian of a launch. A launch carries up to SYRK_MAX_PROBS of them in ONE unit queue (the three K = 4096 inputs of a
// Llama block: one triangular tail instead of three); units of problem p are [unit0, unit0 + S * ntiles_p).
struct SyrkProb {
    int64_t ldx;       // row stride of every sample, elements
    float* part;       // [S * ntiles_p][256*256] fp32, fragment order
    double* dpart;     // [S][nb * 256] fp64: per (chunk, channel) sum of squares from the diagonal tiles (see "exact diagonal")
    int K;
    int nb;            // ceil(K / 256)
    int ntiles_p;      // padded tile count (tiles_padded(nb))
    int S;             // token chunks
    int n;             // samples
    int smp0;          // index of its first sample in SyrkArgs::smp
    int unit0;         // its first unit in the launch's queue
    int pad_;
    uint32_t cb[SYRK_MAX_CHUNKS + 1];   // chunk s = groups [cb[s], cb[s + 1]) of the padded token axis (even boundaries)
    uint32_t ci[SYRK_MAX_CHUNKS + 1];   // the sample (relative to smp0) that holds group cb[s]
};

struct SyrkArgs {
    int P;             // problems
    int nunits;        // units of all problems
    int no_dwave;      // k1_fp32_diag: diagonal tiles are computed in full (rounds 1-5), no fp64 diagonal
    int pad_;
    unsigned* sync;    // round barrier counter (zeroed before the launch), or null
    SyrkProb pr[SYRK_MAX_PROBS];
    SyrkSample smp[SYRK_MAX_SAMPLES];
};

template <int N> __device__ __forceinline__ void dma_wait_upto() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// -----------------------------------------------------------------------------------------------------
// k_syrk4 — one wave per SIMD: 4 waves as 2(M) x 2(N), each wave 128 x 128 = 4 x 4 MFMA 32x32x16 accumulators
// (256 accumulator registers; the kernel uses the whole 512-entry file of its SIMD).
//   * 8 fragment reads feed 16 MFMAs per 16-token slice (0.5 per MFMA);
//   * LDS ring of 4 stages of 32 tokens (128 KiB), 3 stages of LDS-DMA in flight, counted vmcnt (never 0 in the
//     steady state); fragments double-buffered in registers at 16-token granularity; one s_barrier per stage, placed
//     BETWEEN the stage's two MFMA bursts, so every wave arrives with 16 MFMAs' worth of operands already in registers;
//   * the stage loop is unrolled over the ring: slots are compile-time, LDS-DMA destinations are immediates
//     (the kernel owns the whole LDS, base 0), the scalar source offset of each piece lives in its own SGPR and is
//     advanced right after use — a piece is  s_mov m0 / s_nop 0 / buffer_load ... lds / s_add  with no hazard padding;
//   * every MFMA is followed by at most one fragment read (2 ds_read_b64_tr_b16) or one LDS-DMA piece, pinned in
//     that order: with a single wave per SIMD nothing else hides their issue;
//   * two buffer descriptors are live: the current group's and the next group's (a piece requested during group g
//     belongs to group g or g + 1); the pair is rotated once per group from the lane table (see the file header).
// ABL (lab builds only): 1 = no LDS-DMA, 2 = no fragment reads, 4 = LDS-DMA source pinned to the chunk's first stages
// (L2-resident: separates issue cost from miss latency), 8 = pieces without their buffer_load (SALU only),
// 16 = toggling register operands.
// -----------------------------------------------------------------------------------------------------
static constexpr int S4_THREADS = 256;
static constexpr int S4_TOK = 32;
static constexpr int S4_PANEL = S4_TOK * TM * 2;      // 16 KiB
static constexpr int S4_STAGE = 2 * S4_PANEL;         // 32 KiB
static constexpr int NSLOT = 4;
static_assert(NSLOT * S4_TOK == GROUP_TOK, "a group is one turn of the ring");

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// One LDS-DMA piece: LDS destination = immediate DST + the wave's KiB offset (an SGPR; M0 is written by the add and
// is not live across statements in this kernel), source = rsrc.base + soff + voff[lane]; soff then advances by
// `inc` for the piece's next stage — written a whole stage before it is read again, so no hazard padding.
template <int DST, bool LOAD, bool ADVANCE>
__device__ __forceinline__ void dma16w(i32x4 rsrc, uint32_t voff, uint32_t& soff, uint32_t inc, uint32_t wvoff) {
    if constexpr (LOAD && ADVANCE)
        asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\ts_add_u32 %0, %0, %5"
                     : "+s"(soff) : "v"(voff), "s"(rsrc), "s"(wvoff), "n"(DST), "s"(inc) : "memory", "scc");
    else if constexpr (LOAD)
        asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
                     : "+s"(soff) : "v"(voff), "s"(rsrc), "s"(wvoff), "n"(DST) : "memory", "scc");
    else
        asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\ts_add_u32 %0, %0, %3"
                     : "+s"(soff) : "s"(wvoff), "n"(DST), "s"(inc) : "memory", "scc");
}

template <int DT, int ABL>
__global__ __launch_bounds__(S4_THREADS) void k_syrk4(const SyrkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_AS char* lds = (LDS_AS char*)smem;
    if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // DMA destinations are immediates: dynamic LDS must start at 0
    constexpr int PER = 8;                  // LDS-DMA pieces per stage per wave (4 per panel)
    constexpr bool DMA = !(ABL & 1);
    constexpr bool RD = !(ABL & 2);
    constexpr bool ADV = !(ABL & 4);
    constexpr bool LOAD = !(ABL & 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;

    // DMA: piece q (0..3) of wave wv fills KiB-block (q*4 + wv) of a 16-KiB panel = token rows 2*blk, 2*blk + 1
    const int lr = lane >> 5;
    const int c16 = lane & 31;
    const int row_lo = 2 * wv + lr;                       // token row inside the 8-row slab of a piece index q
    const int u_log = (c16 >> 2) ^ (row_lo & 3);          // 8-row slabs keep (row & 3)
    const int ch_off = (u_log * 4 + (c16 & 3)) * 8;

    // fragment addresses: slot j lives at j * 32 KiB; the ds_read offset field reaches 64 KiB, so one base register
    // per PAIR of slots (+ immediate 0 / 32 KiB) covers the ring
    const int p = lane & 15;
    const int trow = 8 * (lane >> 5) + (p >> 2);
    const int sub = 32 * ((lane >> 4) & 1) + 8 * (p & 3);
    constexpr int NPAIR = (NSLOT + 1) / 2;
    int offA[NPAIR][4], offB[NPAIR][4];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            offA[pr][m] = pr * 2 * S4_STAGE + trow * (TM * 2) + (((4 * wm + m) ^ (p >> 2)) << 6) + sub;
            offB[pr][m] = pr * 2 * S4_STAGE + S4_PANEL + trow * (TM * 2) + (((4 * wn + m) ^ (p >> 2)) << 6) + sub;
        }

    const int G = gridDim.x;
    const int lw = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // XCD-contiguous logical id
    const int nunits = a.nunits;
    const int nrounds = (nunits + G - 1) / G;
    const uint32_t wvoff = (uint32_t)wv * 1024u;

    for (int round = 0; round < nrounds; ++round) {
        // Re-align the grid once per round: the workgroups of an XCD share their A/B panels through the XCD's
        // L2 only while they sit at the same token position; without this they drift apart over a ~450-step
        // unit and re-fetch the panels from the fabric (measured: L2 hit 57 %, 7x the algorithmic HBM bytes).
        if (a.sync && round > 0) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)round * (unsigned)G;
                int spins = 0;
                while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 22)) {
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                }
                // a barrier that gave up (another stream holds CUs: some workgroups of this grid are not resident yet) is
                // counted, not silent: sync[1] is read back by llmc_hessian_accum_barrier_timeouts (the result is still
                // correct; the workgroups have lost their common token position and re-fetch their panels from the fabric)
                if (spins >= (1 << 22)) __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        const int u = lw + round * G;
        if (u >= nunits) continue;
        int pi = 0;
#pragma unroll
        for (int q = 1; q < SYRK_MAX_PROBS; ++q)
            if (q < a.P && u >= a.pr[q].unit0) pi = q;
        const SyrkProb& pr = a.pr[pi];
        const int ul = u - pr.unit0;
        const int s = ul / pr.ntiles_p;
        const int ti = ul - s * pr.ntiles_p;
        const TileIdx t = decode_tile(ti, pr.nb);
        if (!t.valid) continue;
        const int64_t row_bytes = pr.ldx * 2;
        const int64_t group_bytes = (int64_t)GROUP_TOK * row_bytes;
        const uint32_t slab = (uint32_t)(8 * row_bytes);
        const uint32_t stage_bytes = (uint32_t)(S4_TOK * row_bytes);
        const uint32_t gb = pr.cb[s];
        const int ngroups = (int)(pr.cb[s + 1] - gb);
        const int i0 = (int)pr.ci[s];
        // Exact diagonal. In a DIAGONAL tile the upper-right 128 x 128 quadrant (wave wm = 0, wn = 1) is the transpose of the
        // lower-left one: k_syrk_fixup mirrors that one instead, and this wave spends its MFMAs on the eight 32 x 32 blocks ON
        // the diagonal (A fragment x the same A fragment, B x the same B), restarting the accumulators every 2048 tokens (FOLD_GROUPS)
        // and folding each block's diagonal into fp64: diag(H) then carries the rounding of 128 chained MFMAs and an
        // fp64 sum over the folds (~2e-8 relative) instead of an fp32 chain over the whole chunk (2-3e-6, twice the
        // reference's sgemm) — diag(H) is what GPTQ's actorder sorts and what the damping averages (gptq.py:63, 169).
        const bool dex = !a.no_dwave && t.bi == t.bj;
        const bool dwave = dex && wv == 1;
        // In such a tile the B panel would be a copy of the A panel: EVERY wave reads its B fragments from the A panel instead
        // (offBu), the B pieces of the three walking waves carry an out-of-range offset (no memory access, zeros into a panel
        // nobody reads) and the diagonal wave does not issue its B pieces at all — half of its LDS-DMA issue slots, which is
        // what pays for its folds: with all of a wave's pieces and fragment reads its instruction stream takes as long as the
        // other waves' MFMA stream, and a round's barrier waits for its slowest unit (profiles/r06_k1_ab.txt). (Moving the
        // diagonal wave's A pieces to a walking wave as well bought nothing more: measured, gpurun_out/r06v.)
        int offBu[NPAIR][4];
#pragma unroll
        for (int pr_ = 0; pr_ < NPAIR; ++pr_)
#pragma unroll
            for (int m = 0; m < 4; ++m) offBu[pr_][m] = offB[pr_][m] - (dex ? S4_PANEL : 0);

        // lane table: descriptor words of sample i0 + lane as this unit sees it (file header, "sample table")
        int v0, v1, v2, vend;
        {
            int li = i0 + lane;
            const bool ok = li < pr.n;
            if (!ok) li = pr.n - 1;
            const SyrkSample e = a.smp[pr.smp0 + li];
            const int64_t srel = ((int64_t)e.g0 - (int64_t)gb) * group_bytes;   // unit-relative offset of its first row
            const uint64_t vb = e.base - (uint64_t)srel;
            int64_t endb = srel + (int64_t)e.T * row_bytes;                     // ... of the end of its last row
            const int64_t unit_end = (int64_t)ngroups * group_bytes;            // nothing past the chunk is ever read: the
            if (endb > unit_end) endb = unit_end;                               // run-ahead requests and the padding group come back as zeros
            const uint32_t nrec = (!ok || endb <= 0) ? 0u : (endb > 0xffffffffll ? 0xffffffffu : (uint32_t)endb);
            v0 = (int)(uint32_t)vb;
            v1 = (int)((uint32_t)(vb >> 32) & 0xffffu);                         // stride 0
            v2 = (int)nrec;
            vend = ok ? (int)((int64_t)e.g0 + (int64_t)((e.T + GROUP_TOK - 1) / GROUP_TOK) - (int64_t)gb) : 0x7fffffff;
        }
        // two descriptor sets that swap roles every group: in an even group dA is the current group's sample and dB the
        // next group's, in an odd group the other way round. The set of group g is dead after the group's first burst
        // and is then rebuilt for group g + 2 from the lane table, its few scalar instructions placed in MFMA gaps that
        // carry neither a fragment read nor a DMA piece (one wave per SIMD: every issue slot outside a gap is lost).
        int rel = __builtin_amdgcn_readfirstlane(0);     // lane of the sample the most recently built descriptor is in
        int cur_end = __builtin_amdgcn_readlane(vend, 0);
        i32x4 dA, dB;
        dA[0] = __builtin_amdgcn_readlane(v0, 0);
        dA[1] = __builtin_amdgcn_readlane(v1, 0);
        dA[2] = __builtin_amdgcn_readlane(v2, 0);
        dA[3] = 0x00020000;
        {
            int rc;
            const int one = __builtin_amdgcn_readfirstlane(1);
            asm volatile("s_cmp_ge_i32 %2, %3\n\ts_addc_u32 %0, %0, 0\n\ts_min_i32 %1, %0, 63"
                         : "+s"(rel), "=s"(rc) : "s"(one), "s"(cur_end) : "scc");
            dB[0] = __builtin_amdgcn_readlane(v0, rc);
            dB[1] = __builtin_amdgcn_readlane(v1, rc);
            dB[2] = __builtin_amdgcn_readlane(v2, rc);
            dB[3] = 0x00020000;
            cur_end = __builtin_amdgcn_readlane(vend, rc);
        }
        // VALU-written SGPRs (v_readlane) must be 5 wait states old before a VMEM instruction reads them as a descriptor;
        // the pieces are inline asm the hazard recogniser cannot see into. The operands tie the pad to the descriptors.
        asm volatile("s_nop 4" : "+s"(dA[0]), "+s"(dA[1]), "+s"(dA[2]), "+s"(dB[0]), "+s"(dB[1]), "+s"(dB[2]));
        // a diagonal tile loads its panel into both LDS panels (vB == vA): the loop below never branches
        const uint32_t vA = (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bi * TM + ch_off) * 2);
        const uint32_t vB = dex ? 0x80000000u : (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bj * TM + ch_off) * 2);

        double dsum[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) dsum[b] = 0.0;
        f32x16 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        if (ngroups > 0) {
            // scalar source offsets (unit-relative), one per piece: sA[q] belongs to the A half, sB[q] to the B half
            uint32_t sA[4], sB[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sA[q] = sB[q] = (uint32_t)q * slab;
            // piece d of the stage in ring slot SL: d = 0..3 A-panel KiB-block (d*4 + wv), d = 4..7 the same of B;
            // `rs`: the descriptor of the sample the stage's group lies in
            auto piece = [&](auto slc, auto dc, const i32x4& rs) {
                constexpr int SL = decltype(slc)::value;
                constexpr int d = decltype(dc)::value;
                if constexpr (DMA) {
                    constexpr int DSTB = SL * S4_STAGE + (d >> 2) * S4_PANEL + (d & 3) * 4 * 1024;
                    if constexpr (d < 4) dma16w<DSTB, LOAD, ADV>(rs, vA, sA[d & 3], stage_bytes, wvoff);
                    else dma16w<DSTB, LOAD, ADV>(rs, vB, sB[d & 3], stage_bytes, wvoff);
                }
            };
            s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
            if constexpr (!RD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s16x8 v, w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = (short)(lane * 8 + e + i);
                        // TOGGLE: bf16 bit patterns with random sign / mantissa / a few exponent bits, different in
                        // every register set, so that consecutive MFMAs see changing operands without any LDS read
                        uint32_t hsh = (uint32_t)(lane * 977 + e * 131 + i * 7919 + 12345) * 2654435761u;
                        w[e] = (short)((hsh >> 16 & 0x83ff) | 0x3c00 | ((hsh >> 3) & 0x0300));
                    }
                    asm volatile("" : "+v"(v), "+v"(w));
                    fa0[i] = fb0[i] = fa1[i] = fb1[i] = v;
                    if constexpr (ABL & 16) {
                        fa0[i] = w;
                        fb0[i] = w ^ (short)0x1234;
                        fa1[i] = w ^ (short)0x0777;
                        fb1[i] = w ^ (short)0x4321;
                        asm volatile("" : "+v"(fa0[i]), "+v"(fb0[i]), "+v"(fa1[i]), "+v"(fb1[i]));
                    }
                }
            }
            // fragment f of slice kk of the stage in slot SL: order A0 B0 B1 B2 B3 A1 A2 A3 (the serpentine walk
            // starts every burst at (0, 0) and needs row m's A fragment from step 4m on)
            auto frag = [&](auto slc, int kk, auto fc, s16x8 (&fa)[4], s16x8 (&fb)[4]) {
                constexpr int SL = decltype(slc)::value;
                constexpr int f = decltype(fc)::value;
                if constexpr (RD) {
                    constexpr int IMM = (SL & 1) * S4_STAGE;
                    if constexpr (f == 0) fa[0] = tr_frag(lds + offA[SL >> 1][0], IMM + kk * 16 * TM * 2);
                    else if constexpr (f <= 4) fb[f - 1] = tr_frag(lds + offBu[SL >> 1][f - 1], IMM + kk * 16 * TM * 2);
                    else fa[f - 4] = tr_frag(lds + offA[SL >> 1][f - 4], IMM + kk * 16 * TM * 2);
                }
            };
            // one burst: 16 MFMAs on (fa, fb). Behind MFMA i: i in {0,1,2,4,5,6,8,9} -> the next fragment of slice kk
            // of slot RSL into (na, nb) (all eight are back six MFMAs before the burst ends); i in {3,7,11,15} ->
            // LDS-DMA piece D0 + i/4 of slot DSL (one KiB per wave every four MFMAs = 32 B/clk per CU, evenly spread);
            // i in {10,12,13,14} -> `extra(i)`: scalar bookkeeping that must not cost an issue slot of its own
            // DW (the diagonal wave of a diagonal tile): 0 = the 4x4 block walk, 1 = the eight blocks on the diagonal. The diagonal
            // blocks take slots whose operand is already
            // in registers in the walk's fragment order: 0..3 -> fb[i] x fb[i], 4 -> fa[1], 5 -> fa[0], 8 -> fa[2], 12 -> fa[3];
            // every slot keeps its fragment read / DMA piece (the wave still stages its share of the tile).
            auto burst = [&](auto dwc, const s16x8 (&fa)[4], const s16x8 (&fb)[4], s16x8 (&na)[4], s16x8 (&nb)[4], auto rslc, int rd_kk,
                             auto dslc, auto d0c, const i32x4& rs, auto&& extra) {
                constexpr int D0 = decltype(d0c)::value;
                constexpr int DW = decltype(dwc)::value;
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // serpentine walk of the 4x4 accumulator block: one operand changes per step
                    constexpr int mi = i >> 2, ni = (mi & 1) ? 3 - (i & 3) : (i & 3);
                    if constexpr (DW == 0) {
                        acc[mi][ni] = Mfma<DT>::run(fa[mi], fb[ni], acc[mi][ni]);
                    } else {
                        if constexpr (i < 4) acc[0][i] = Mfma<DT>::run(fb[i], fb[i], acc[0][i]);
                        else if constexpr (i == 4) acc[1][0] = Mfma<DT>::run(fa[1], fa[1], acc[1][0]);
                        else if constexpr (i == 5) acc[1][1] = Mfma<DT>::run(fa[0], fa[0], acc[1][1]);
                        else if constexpr (i == 8) acc[2][0] = Mfma<DT>::run(fa[2], fa[2], acc[2][0]);
                        else if constexpr (i == 12) acc[3][0] = Mfma<DT>::run(fa[3], fa[3], acc[3][0]);
                    }
                    if constexpr ((i & 3) == 3) {
                        if constexpr (!(DW != 0 && D0 == 4)) piece(dslc, std::integral_constant<int, D0 + (i >> 2)>{}, rs);   // the diagonal wave: no B pieces
                    }
                    else if constexpr (i < 10) frag(rslc, rd_kk, std::integral_constant<int, i - (i >> 2)>{}, na, nb);
                    else extra(ic);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            auto nothing = [](auto) {};
            // one group (ring turn) g: `cur` = descriptor of group g, `nxt` = of group g + 1; `cur` is rebuilt for group
            // g + 2 behind the second burst of the group's first stage
            auto group = [&](auto dwg, i32x4& cur, const i32x4& nxt, int g) {
                constexpr int DWG = decltype(dwg)::value;               // 0: the block walk, 1: the diagonal wave (see burst)
                static_for<0, NSLOT>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;              // stage st = g*NSLOT + J sits in slot J
                    constexpr int JN = (J + 1) % NSLOT;                 // slot of stage st+1
                    constexpr int JP = (J + NSLOT - 1) % NSLOT;         // slot of stage st+NSLOT-1 (= st-1)
                    constexpr auto dw1 = std::integral_constant<int, DWG>{};
                    constexpr auto dw0 = dw1;
                    // slice 0 of stage st; fetches slice 1; requests the B half of stage st+NSLOT-1 (J = 0: the last
                    // stage of this group, otherwise a stage of the next group)
                    burst(dw0, fa0, fb0, fa1, fb1, jc, 1, std::integral_constant<int, JP>{}, std::integral_constant<int, 4>{},
                          J == 0 ? cur : nxt, nothing);
                    // this wave's pieces of stage st+1 have landed (NSLOT-2 later stages may still be in flight);
                    // every wave has read the whole of stage st once its lgkmcnt(0) is behind the barrier
                    dma_wait_upto<(NSLOT - 2) * (DWG ? PER / 2 : PER)>();
                    lds_wait_all();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // slice 1; fetches slice 0 of stage st+1; requests the A half of stage st+NSLOT (next group) into slot J
                    if constexpr (J == 0) {
                        int rc = 0;
                        burst(dw1, fa1, fb1, fa0, fb0, std::integral_constant<int, JN>{}, 0, jc, std::integral_constant<int, 0>{}, nxt,
                              [&](auto ic) {
                                  constexpr int i = decltype(ic)::value;
                                  if constexpr (i == 10) {      // group g + 2 starts a new sample iff it lies at or past the end
                                      const int g2 = g + 2;     // of the sample group g + 1 is in. Three SALU instructions
                                      asm volatile("s_cmp_ge_i32 %2, %3\n\ts_addc_u32 %0, %0, 0\n\ts_min_i32 %1, %0, 63"
                                                   : "+s"(rel), "=s"(rc) : "s"(g2), "s"(cur_end) : "scc");
                                  } else if constexpr (i == 12) {
                                      cur[0] = __builtin_amdgcn_readlane(v0, rc);
                                      cur[1] = __builtin_amdgcn_readlane(v1, rc);
                                  } else if constexpr (i == 13) {
                                      cur[2] = __builtin_amdgcn_readlane(v2, rc);
                                      cur_end = __builtin_amdgcn_readlane(vend, rc);
                                  }
                              });
                    } else {
                        burst(dw1, fa1, fb1, fa0, fb0, std::integral_constant<int, JN>{}, 0, jc, std::integral_constant<int, 0>{}, nxt,
                              nothing);
                    }
                });
            };
            // The diagonal wave's eight block diagonals into fp64, every FOLD_GROUPS groups (2048 tokens: 128 chained MFMAs per
            // block between folds — their fp32 rounding, ~2e-7, averages down over the folds of a chunk and the chunks of a
            // launch; the final fp32 rounding of diag(H) dominates). Then the eight accumulators restart from zero. A fold is
            // ~2000 cycles of this wave's time: per group it must stay well inside the slack its halved MFMA load leaves, or
            // every round (whose barrier waits for its slowest unit) pays for it — folding every pair of groups cost 3 ms per
            // bench step (profiles/r06_k1_ab.txt).
            auto fold = [&]() {
                    // Lane l of a 32x32 accumulator holds column j = l & 31 and rows 8 q + 4 (l >> 5) + r in register
                // 4 q + r: element (j, j) sits in register k = (j & 3) + 4 (j >> 3) of lane j + 32 ((j >> 2) & 1) — every
                // register carries the diagonal in exactly two lanes, so sixteen accumulator reads under a two-lane EXEC
                // mask each collect the block's diagonal into one VGPR (no per-lane select chain, no mask registers).
                // s_nop: an MFMA result must be 11+ wait states old before a VALU reads it (invisible to the compiler in asm).
                auto dg = [&](const f32x16& c) -> float {
                    float t = 0.0f;
                    uint64_t save;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b32 exec_lo, 0x00000001\n\ts_mov_b32 exec_hi, 0x00000010\n\tv_accvgpr_read_b32 %[t], %[c0]\n\t"
                        "s_mov_b32 exec_lo, 0x00000002\n\ts_mov_b32 exec_hi, 0x00000020\n\tv_accvgpr_read_b32 %[t], %[c1]\n\t"
                        "s_mov_b32 exec_lo, 0x00000004\n\ts_mov_b32 exec_hi, 0x00000040\n\tv_accvgpr_read_b32 %[t], %[c2]\n\t"
                        "s_mov_b32 exec_lo, 0x00000008\n\ts_mov_b32 exec_hi, 0x00000080\n\tv_accvgpr_read_b32 %[t], %[c3]\n\t"
                        "s_mov_b32 exec_lo, 0x00000100\n\ts_mov_b32 exec_hi, 0x00001000\n\tv_accvgpr_read_b32 %[t], %[c4]\n\t"
                        "s_mov_b32 exec_lo, 0x00000200\n\ts_mov_b32 exec_hi, 0x00002000\n\tv_accvgpr_read_b32 %[t], %[c5]\n\t"
                        "s_mov_b32 exec_lo, 0x00000400\n\ts_mov_b32 exec_hi, 0x00004000\n\tv_accvgpr_read_b32 %[t], %[c6]\n\t"
                        "s_mov_b32 exec_lo, 0x00000800\n\ts_mov_b32 exec_hi, 0x00008000\n\tv_accvgpr_read_b32 %[t], %[c7]\n\t"
                        "s_mov_b32 exec_lo, 0x00010000\n\ts_mov_b32 exec_hi, 0x00100000\n\tv_accvgpr_read_b32 %[t], %[c8]\n\t"
                        "s_mov_b32 exec_lo, 0x00020000\n\ts_mov_b32 exec_hi, 0x00200000\n\tv_accvgpr_read_b32 %[t], %[c9]\n\t"
                        "s_mov_b32 exec_lo, 0x00040000\n\ts_mov_b32 exec_hi, 0x00400000\n\tv_accvgpr_read_b32 %[t], %[c10]\n\t"
                        "s_mov_b32 exec_lo, 0x00080000\n\ts_mov_b32 exec_hi, 0x00800000\n\tv_accvgpr_read_b32 %[t], %[c11]\n\t"
                        "s_mov_b32 exec_lo, 0x01000000\n\ts_mov_b32 exec_hi, 0x10000000\n\tv_accvgpr_read_b32 %[t], %[c12]\n\t"
                        "s_mov_b32 exec_lo, 0x02000000\n\ts_mov_b32 exec_hi, 0x20000000\n\tv_accvgpr_read_b32 %[t], %[c13]\n\t"
                        "s_mov_b32 exec_lo, 0x04000000\n\ts_mov_b32 exec_hi, 0x40000000\n\tv_accvgpr_read_b32 %[t], %[c14]\n\t"
                        "s_mov_b32 exec_lo, 0x08000000\n\ts_mov_b32 exec_hi, 0x80000000\n\tv_accvgpr_read_b32 %[t], %[c15]\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [t] "+v"(t), [sv] "=&s"(save)
                        : [c0] "a"(c[0]), [c1] "a"(c[1]), [c2] "a"(c[2]), [c3] "a"(c[3]), [c4] "a"(c[4]), [c5] "a"(c[5]),
                          [c6] "a"(c[6]), [c7] "a"(c[7]), [c8] "a"(c[8]), [c9] "a"(c[9]), [c10] "a"(c[10]), [c11] "a"(c[11]),
                          [c12] "a"(c[12]), [c13] "a"(c[13]), [c14] "a"(c[14]), [c15] "a"(c[15]));
                    return t;
                };
                asm volatile("s_nop 15" ::: "memory");      // the last MFMA of the burst is at most a few instructions old
                dsum[0] += (double)dg(acc[0][0]);
                dsum[1] += (double)dg(acc[0][1]);
                dsum[2] += (double)dg(acc[0][2]);
                dsum[3] += (double)dg(acc[0][3]);
                dsum[4] += (double)dg(acc[1][0]);
                dsum[5] += (double)dg(acc[1][1]);
                dsum[6] += (double)dg(acc[2][0]);
                dsum[7] += (double)dg(acc[3][0]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[0][0][r] = 0.0f; acc[0][1][r] = 0.0f; acc[0][2][r] = 0.0f; acc[0][3][r] = 0.0f;
                    acc[1][0][r] = 0.0f; acc[1][1][r] = 0.0f; acc[2][0][r] = 0.0f; acc[3][0][r] = 0.0f;
                }
            };
            // prologue: stages 0 .. NSLOT-2 and the A half of stage NSLOT-1 (all of group 0) requested, stage 0 published
            if (dwave) {      // A halves only (see offBu)
                static_for<0, NSLOT>([&](auto slc) { static_for<0, 4>([&](auto dc) { piece(slc, dc, dA); }); });
                dma_wait_upto<(NSLOT - 2) * (PER / 2) + PER / 2>();
            } else {
                static_for<0, NSLOT - 1>([&](auto slc) { static_for<0, 8>([&](auto dc) { piece(slc, dc, dA); }); });
                static_for<0, 4>([&](auto dc) { piece(std::integral_constant<int, NSLOT - 1>{}, dc, dA); });
                dma_wait_upto<(NSLOT - 2) * PER + PER / 2>();
            }
            __builtin_amdgcn_s_barrier();
            static_for<0, 8>([&](auto fc) { frag(std::integral_constant<int, 0>{}, 0, fc, fa0, fb0); });
            // groups in pairs (the descriptor sets swap roles); an odd chunk gets one padding group of zeros
            if (dwave) {
                // long units fold every FOLD_GROUPS groups; short ones (small calibration sets: few folds to average over) every pair
                const int fmask = ngroups >= 8 * FOLD_GROUPS ? FOLD_GROUPS - 1 : 1;
                for (int g = 0; g < ngroups; g += 2) {
                    group(std::integral_constant<int, 1>{}, dA, dB, g);
                    group(std::integral_constant<int, 1>{}, dB, dA, g + 1);
                    if (((g + 2) & fmask) == 0 || g + 2 >= ngroups) fold();
                }
            } else {
                for (int g = 0; g < ngroups; g += 2) {
                    group(std::integral_constant<int, 0>{}, dA, dB, g);
                    group(std::integral_constant<int, 0>{}, dB, dA, g + 1);
                }
            }
            lds_wait_all();                  // the trailing fragment reads
            __builtin_amdgcn_s_barrier();    // ... of every wave, before the next unit's prologue overwrites the ring
        }

        // partial tile in fragment order of a 2x4 wave grid with 4x2 accumulators per wave (what k_syrk_fixup decodes):
        // wave (wm, wn) x accumulator (m, n) of the 2x2 / 4x4 layout is wave (wm, 2*wn + n/2) x accumulator (m, n%2)
        if (dwave) {
            // sum over this chunk's tokens of x^2 for the tile's 256 channels: block -> channels as the fragments were read
            // (fb[n]: 128 + 32 n .., fa[m]: 32 m ..); the quadrant itself is mirrored from the lower-left one by k_syrk_fixup
            const int dj = lane & 31;
            if ((((dj >> 2) & 1) == (lane >> 5))) {                  // the lanes that hold a diagonal element
                double* dp = pr.dpart + ((int64_t)s * pr.nb + t.bi) * TM + dj;
                dp[128] = dsum[0];
                dp[160] = dsum[1];
                dp[192] = dsum[2];
                dp[224] = dsum[3];
                dp[32] = dsum[4];
                dp[0] = dsum[5];
                dp[64] = dsum[6];
                dp[96] = dsum[7];
            }
        } else {
            float* slot = pr.part + (int64_t)ul * TILE_FLOATS;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2],
                                   acc[m][n][4 * q + 3]};
                        const int wv8 = wm * 4 + wn * 2 + (n >> 1);
                        int idx = ((((wv8 * 4 + m) * 2 + (n & 1)) * 4 + q) * 64 + lane);
                        *reinterpret_cast<f32x4*>(slot + (int64_t)idx * 4) = v;
                    }
        }
        // the stores share the VM counter with the next unit's LDS-DMA: drain them (and the trailing requests)
        dma_wait_all();
    }
}

// Sum the S partials of each tile (chunk order), H <- alpha*H + beta*sum, mirror to the upper triangle. One thread per float4
// of the fragment-order tile: idx -> (wv,m,n,q,lane) -> rows i0..i0+3, column j. grid.z = problem.
// Diagonal tiles: their upper-right quadrant was not computed (its wave formed the exact diagonal instead): the lower-left
// quadrant is mirrored into it like an off-diagonal tile. skip_diag: the diagonal entries are left to k_diag_apply.
struct FixupProb {
    const float* part;
    float* H;
    int K, nb, ntiles_p, S;
    float alpha, beta;
};
struct FixupArgs {
    FixupProb pr[SYRK_MAX_PROBS];
    int exact;         // 1: diagonal tiles carry no upper-right quadrant (mirrored here) and diag(H) is left to k_diag_apply
};

__global__ __launch_bounds__(256) void k_syrk_fixup(const FixupArgs fa) {
    const FixupProb& a = fa.pr[blockIdx.z];
    const int ti = blockIdx.y;
    if (ti >= a.ntiles_p) return;
    const TileIdx t = decode_tile(ti, a.nb);
    if (!t.valid) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;  // 0 .. 16383
    const int lane = idx & 63;
    const int q = (idx >> 6) & 3;
    const int n = (idx >> 8) & 1;
    const int m = (idx >> 9) & 3;
    const int wv = idx >> 11;
    const int wm = wv >> 2, wn = wv & 3;
    const bool dtile = fa.exact && t.bi == t.bj;
    if (dtile && wm == 0 && wn >= 2) return;         // mirrored from the lower-left quadrant below
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.S; ++s) {
        const float* slot = a.part + ((int64_t)s * a.ntiles_p + ti) * TILE_FLOATS;
        f32x4 v = *reinterpret_cast<const f32x4*>(slot + (int64_t)idx * 4);
        sum += v;
    }
    const int K = a.K;
    float* H = a.H;
    const int i0 = t.bi * TM + wm * 128 + m * 32 + 8 * q + 4 * (lane >> 5);
    const int j = t.bj * TM + wn * 64 + n * 32 + (lane & 31);
    if (j >= K) return;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = i0 + r;
        if (i < K) {
            float h = a.beta * sum[r];
            if (a.alpha != 0.0f) h += a.alpha * H[(int64_t)i * K + j];
            o[r] = h;
            if (!(fa.exact && i == j)) H[(int64_t)i * K + j] = h;
        }
    }
    if (t.bi != t.bj || (dtile && wm == 1 && wn < 2)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i0 + r < K) H[(int64_t)j * K + i0 + r] = o[r];
    }
}

// Exact diagonal: d[j] = (n_before / n_after) d_old[j] + (2 / n_after) * sum over chunks of dpart[s][j] in fp64, H[j][j] = (float) d[j].
// d_old is the caller's fp64 running diagonal (dstate, then updated) or, without one, H's own fp32 diagonal (which k_syrk_fixup
// left untouched): half an ulp of extra rounding per launch, nothing when a Hessian is accumulated in one launch.
struct DiagProb {
    const double* dpart;
    double* dstate;
    float* H;
    int K, nbK, S, pad_;
    double alpha, beta;
};
struct DiagArgs {
    DiagProb pr[SYRK_MAX_PROBS];
};

__global__ __launch_bounds__(256) void k_diag_apply(const DiagArgs da) {
    const DiagProb& a = da.pr[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.K) return;
    double s = 0.0;
    for (int i = 0; i < a.S; ++i) s += a.dpart[(int64_t)i * a.nbK + j];
    double d = a.beta * s;
    if (a.alpha != 0.0) d += a.alpha * (a.dstate ? a.dstate[j] : (double)a.H[(int64_t)j * a.K + j]);
    if (a.dstate) a.dstate[j] = d;
    a.H[(int64_t)j * a.K + j] = (float)d;
}

static inline double chunk_cost(int64_t units, int64_t ngroups, int S, int ncu) {
    // a simple time model (microseconds): rounds * (groups per unit * t_group + t_unit) + fixup traffic (S partial tiles
    // written + read)
    const double t_group = 1.8, t_unit = 6.0, fix_us_per_tile = 0.13;  // 2 x 256 KiB at ~4 TB/s
    const double rounds = (double)ceil_div64(units, ncu);
    return rounds * ((double)ceil_div64(ngroups, S) * t_group + t_unit) + fix_us_per_tile * (double)units;
}

// pick S >= Smin minimising the model; `copies` identical problems share the queue (their units fill rounds together)
static inline int choose_chunks(int ntiles_real, int64_t ngroups, int64_t x_bytes, int ncu, int copies) {
    int smin = (int)(x_bytes / (1ll << 31)) + 1;   // unit-relative byte offsets stay below 2^31
    if (smin > ngroups) smin = (int)ngroups;
    if (smin < 1) smin = 1;
    int best = smin;
    double best_cost = 1e30;
    for (int S = smin; S <= SYRK_MAX_CHUNKS && S <= ngroups; ++S) {
        const double cost = chunk_cost((int64_t)ntiles_real * S * copies, ngroups, S, ncu);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = S;
        }
    }
    return best;
}

}  // namespace llmc

using namespace llmc;

// Geometry of one problem: the padded token axis, the chunks, and which sample each chunk starts in. Fills the geometry
// fields of `p` (not the pointers) and its samples' T / g0 in smp[0 .. n). Returns 0, or LLMC_EINVAL with the message set.
static int syrk_plan(const int64_t* T_list, int n, int64_t K, int64_t ldx, int copies, SyrkProb* p, SyrkSample* smp) {
    LLMC_REQUIRE(T_list && n >= 1, "hessian_accum: empty sample list");
    LLMC_REQUIRE(K > 0 && K < (1 << 30) && ldx >= K && ldx % 8 == 0, "hessian_accum: rows must be 16-B aligned, K < 2^30");
    int64_t G = 0;
    for (int i = 0; i < n; ++i) {
        LLMC_REQUIRE(T_list[i] > 0 && T_list[i] < (1ll << 31), "hessian_accum: every sample needs 0 < tokens < 2^31");
        smp[i].T = (uint32_t)T_list[i];
        smp[i].g0 = (uint32_t)G;
        G += ceil_div64(T_list[i], GROUP_TOK);
    }
    LLMC_REQUIRE(G < (1ll << 31), "hessian_accum: too many tokens in one call");
    p->ldx = ldx;
    p->K = (int)K;
    p->nb = (int)ceil_div64(K, TM);
    p->ntiles_p = tiles_padded(p->nb);
    p->n = n;
    p->pad_ = 0;
    const int real = p->nb * (p->nb + 1) / 2;
    int S = choose_chunks(real, G, G * GROUP_TOK * ldx * 2, 256, copies);
#ifdef LLMC_LAB
    if (const char* e = lab_env("LLMC_SYRK_S")) S = atoi(e);   // lab: force the token-chunk count
#endif
    if (S > G / 2) S = (int)(G / 2);     // every chunk at least one group pair
    if (S < 1) S = 1;
    for (;; ++S) {
        LLMC_REQUIRE(S <= SYRK_MAX_CHUNKS && (S <= G / 2 || S == 1), "hessian_accum: samples too short for one call (stage them into fewer, longer ones)");
        int i = 0;
        bool fits = true;
        for (int s = 0; s <= S; ++s) {
            const int64_t g = s == S ? G : (s * G / S) & ~(int64_t)1;   // interior boundaries even: the kernel walks group pairs
            p->cb[s] = (uint32_t)g;
            while (i + 1 < n && (int64_t)smp[i + 1].g0 <= g) ++i;
            p->ci[s] = (uint32_t)i;
            // a unit walks the samples ci[s] .. (the one holding group cb[s + 1], whose first stages it requests)
            if (s > 0 && (int)p->ci[s] - (int)p->ci[s - 1] + 1 > SYRK_UNIT_SAMPLES) fits = false;
        }
        if (fits) break;
    }
    p->S = S;
    return LLMC_OK;
}

static inline size_t align256z(size_t x) { return (x + 255) & ~(size_t)255; }

// The whole launch: every problem's geometry, its slices of the workspace (partial tiles, fp64 diagonal partials) and the
// round-barrier words behind them. ws may be null (sizes only). Problems with the same shape share the chunk-count search.
static int syrk_plan_multi(const llmc_hessian_problem_t* probs, int P, void* ws, SyrkArgs* a, size_t* total) {
    LLMC_REQUIRE(probs && P >= 1 && P <= SYRK_MAX_PROBS, "hessian_accum_multi: 1 .. LLMC_HESSIAN_MAX_PROBLEMS problems per call");
    int nsmp = 0, unit0 = 0;
    size_t off = 0;
    for (int q = 0; q < P; ++q) {
        const llmc_hessian_problem_t& h = probs[q];
        LLMC_REQUIRE(h.n >= 1 && nsmp + h.n <= SYRK_MAX_SAMPLES, "hessian_accum: more than LLMC_HESSIAN_MAX_SAMPLES samples in one call");
        int copies = 0;      // problems of the same shape (K and token lists) fill rounds together
        for (int r = 0; r < P; ++r) {
            bool same = probs[r].K == h.K && probs[r].n == h.n && probs[r].ldx == h.ldx;
            for (int i = 0; same && i < h.n; ++i) same = probs[r].T_list_host && h.T_list_host && probs[r].T_list_host[i] == h.T_list_host[i];
            copies += same ? 1 : 0;
        }
        SyrkProb* p = &a->pr[q];
        int rc = syrk_plan(h.T_list_host, h.n, h.K, h.ldx, copies < 1 ? 1 : copies, p, a->smp + nsmp);
        if (rc) return rc;
        p->smp0 = nsmp;
        p->unit0 = unit0;
        nsmp += h.n;
        unit0 += p->S * p->ntiles_p;
        p->part = ws ? (float*)((char*)ws + off) : nullptr;
        off += align256z((size_t)p->S * p->ntiles_p * TILE_FLOATS * sizeof(float));
        p->dpart = ws ? (double*)((char*)ws + off) : nullptr;
        off += align256z((size_t)p->S * p->nb * TM * sizeof(double));
    }
    a->P = P;
    a->nunits = unit0;
    a->no_dwave = opt(OPT_K1_FP32_DIAG) ? 1 : 0;
    a->pad_ = 0;
    a->sync = ws ? (unsigned*)((char*)ws + off) : nullptr;
    off += 256;
    if (total) *total = off;
    return LLMC_OK;
}

extern "C" int llmc_hessian_max_samples(void) { return SYRK_MAX_SAMPLES; }
extern "C" int llmc_hessian_max_problems(void) { return SYRK_MAX_PROBS; }

extern "C" size_t llmc_hessian_accum_multi_ws_bytes(const llmc_hessian_problem_t* probs_host, int P) {
    SyrkArgs a;
    size_t total = 0;
    if (syrk_plan_multi(probs_host, P, nullptr, &a, &total)) return 0;
    return total;
}

static int syrk_launch(const SyrkArgs& a, int dt, hipStream_t st) {
    int abl = 0;
    SyrkArgs b = a;
#ifdef LLMC_LAB
    if (lab_env("LLMC_SYRK_NOSYNC")) b.sync = nullptr;
    if (const char* e = lab_env("LLMC_SYRK_ABL")) abl = atoi(e);   // wrong results by design
#endif
    // persistent: one workgroup per CU (a multiple of 8 keeps XCDs contiguous), minus the CUs the caller keeps free for
    // kernels of other streams (llmc_hip_set_cu_reserve): a k_syrk4 workgroup owns its CU, nothing co-resides with it
    int grid = (device_cu_count() - cu_reserve()) & ~7;
    if (grid < 8) grid = 8;
    const bool bf = dt == LLMC_BF16;
    const void* fn = bf ? (const void*)k_syrk4<LLMC_BF16, 0> : (const void*)k_syrk4<LLMC_F16, 0>;
#ifdef LLMC_LAB
    if (bf) {
        if (abl == 1) fn = (const void*)k_syrk4<LLMC_BF16, 1>;
        else if (abl == 2) fn = (const void*)k_syrk4<LLMC_BF16, 2>;
        else if (abl == 3) fn = (const void*)k_syrk4<LLMC_BF16, 3>;
        else if (abl == 4) fn = (const void*)k_syrk4<LLMC_BF16, 4>;
        else if (abl == 8) fn = (const void*)k_syrk4<LLMC_BF16, 8>;
        else if (abl == 18) fn = (const void*)k_syrk4<LLMC_BF16, 18>;
        else if (abl == 19) fn = (const void*)k_syrk4<LLMC_BF16, 19>;
    }
#endif
    (void)abl;
    const int lds_bytes = NSLOT * S4_STAGE;
    int rc = ensure_dynamic_lds(fn, lds_bytes);
    if (rc) return rc;
    void* kargs[] = {(void*)&b};
    LLMC_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(S4_THREADS), kargs, (size_t)lds_bytes, st));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

// One launch for the partial tiles of every problem (one unit queue: the problems' triangular tails fill rounds together).
// k1_batch_off (llmc_hip_set_option): one launch per problem instead — same units, same bits.
extern "C" int llmc_hessian_accum_multi_partials(const llmc_hessian_problem_t* probs_host, int P, int dt, void* ws,
                                                 llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "hessian_accum: X must be f16 or bf16");
    LLMC_REQUIRE(ws && ((uintptr_t)ws & 255) == 0, "hessian_accum: workspace must be 256-B aligned");
    SyrkArgs a;
    int rc = syrk_plan_multi(probs_host, P, ws, &a, nullptr);
    if (rc) return rc;
    for (int q = 0; q < P; ++q) {
        LLMC_REQUIRE(probs_host[q].X_list_host, "hessian_accum: null sample list");
        for (int i = 0; i < probs_host[q].n; ++i) {
            const void* x = probs_host[q].X_list_host[i];
            LLMC_REQUIRE(x && ((uintptr_t)x & 15) == 0, "hessian_accum: X rows must be 16-B aligned");
            a.smp[a.pr[q].smp0 + i].base = (uint64_t)(uintptr_t)x;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    LLMC_HIP_CHECK(hipMemsetAsync(a.sync, 0, 8 * SYRK_MAX_PROBS, st));     // per launch: round counter + time-out counter
    if (P > 1 && opt(OPT_K1_BATCH_OFF)) {
        for (int q = 0; q < P; ++q) {
            SyrkArgs one = a;
            one.P = 1;
            one.pr[0] = a.pr[q];
            one.pr[0].unit0 = 0;
            one.nunits = a.pr[q].S * a.pr[q].ntiles_p;
            one.sync = a.sync + 2 * q;
            rc = syrk_launch(one, dt, st);
            if (rc) return rc;
        }
        return LLMC_OK;
    }
    return syrk_launch(a, dt, st);
}

// The ordered reduction of every problem's partial tiles into its H (running mean weights from n_before / n_after) and the
// exact diagonal (k_diag_apply). k1_fp32_diag (llmc_hip_set_option, set for BOTH calls of a pair): the kernel of rounds 1-5 —
// diagonal tiles computed in full, the fp32 chain's own diagonal — for A/B.
extern "C" int llmc_hessian_accum_multi_reduce(const llmc_hessian_problem_t* probs_host, int P, const void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(ws, "hessian_accum_reduce: null workspace");
    SyrkArgs a;
    int rc = syrk_plan_multi(probs_host, P, (void*)ws, &a, nullptr);
    if (rc) return rc;
    FixupArgs f;
    DiagArgs d;
    const bool exact = !opt(OPT_K1_FP32_DIAG);
    int max_tiles = 0, max_k = 0;
    for (int q = 0; q < P; ++q) {
        const llmc_hessian_problem_t& h = probs_host[q];
        LLMC_REQUIRE(h.H && h.n_after > 0, "hessian_accum_reduce: bad argument");
        const SyrkProb& p = a.pr[q];
        f.pr[q].part = p.part; f.pr[q].H = h.H; f.pr[q].K = p.K; f.pr[q].nb = p.nb; f.pr[q].ntiles_p = p.ntiles_p; f.pr[q].S = p.S;
        f.pr[q].alpha = (float)(h.n_before / h.n_after);
        f.pr[q].beta = (float)(2.0 / h.n_after);
        d.pr[q].dpart = p.dpart; d.pr[q].dstate = h.dstate; d.pr[q].H = h.H; d.pr[q].K = p.K; d.pr[q].nbK = p.nb * TM; d.pr[q].S = p.S;
        d.pr[q].pad_ = 0;
        d.pr[q].alpha = h.n_before / h.n_after;
        d.pr[q].beta = 2.0 / h.n_after;
        if (p.ntiles_p > max_tiles) max_tiles = p.ntiles_p;
        if (p.K > max_k) max_k = p.K;
    }
    f.exact = exact ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_syrk_fixup, dim3(TILE_FLOATS / 4 / 256, max_tiles, P), dim3(256), 0, st, f);
    LLMC_LAUNCH_CHECK();
    if (exact) {
        hipLaunchKernelGGL(k_diag_apply, dim3((unsigned)ceil_div64(max_k, 256), P), dim3(256), 0, st, d);
        LLMC_LAUNCH_CHECK();
    }
    return LLMC_OK;
}

// Number of round barriers of the LAST launch on this workspace that gave up waiting (device word behind the partials):
// copies it to *out_host after synchronising `stream`. 0 in a healthy run; > 0 means another stream kept workgroups of the
// persistent grid off their CUs (results stay correct, the kernel re-fetches its panels: slower).
extern "C" int llmc_hessian_accum_multi_barrier_timeouts(const llmc_hessian_problem_t* probs_host, int P, const void* ws,
                                                         unsigned* out_host, llmc_stream_t stream) {
    LLMC_REQUIRE(ws && out_host, "hessian_accum_barrier_timeouts: null argument");
    SyrkArgs a;
    int rc = syrk_plan_multi(probs_host, P, (void*)ws, &a, nullptr);
    if (rc) return rc;
    unsigned w[2 * SYRK_MAX_PROBS] = {};
    LLMC_HIP_CHECK(hipMemcpyAsync(w, a.sync, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream));
    LLMC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    unsigned t = 0;
    for (int q = 0; q < SYRK_MAX_PROBS; ++q) t += w[2 * q + 1];
    *out_host = t;
    return LLMC_OK;
}

// ---- one problem: the entry points of rounds 1-5, now thin wrappers -------------------------------------------------------
static llmc_hessian_problem_t one_problem(float* H, const void* const* X, const int64_t* T, int n, int64_t K, int64_t ldx,
                                          double n_before, double n_after) {
    llmc_hessian_problem_t h;
    h.H = H; h.dstate = nullptr; h.X_list_host = X; h.T_list_host = T; h.n = n; h.K = K; h.ldx = ldx;
    h.n_before = n_before; h.n_after = n_after;
    return h;
}

extern "C" size_t llmc_hessian_accum_ptrs_ws_bytes(const int64_t* T_list_host, int n, int64_t K, int64_t ldx) {
    const llmc_hessian_problem_t h = one_problem(nullptr, nullptr, T_list_host, n, K, ldx, 0.0, 1.0);
    return llmc_hessian_accum_multi_ws_bytes(&h, 1);
}

extern "C" size_t llmc_hessian_accum_ws_bytes(int64_t T, int64_t K, int64_t ldx) {
    return llmc_hessian_accum_ptrs_ws_bytes(&T, 1, K, ldx);
}

extern "C" int llmc_hessian_accum_ptrs_partials(const void* const* X_list_host, const int64_t* T_list_host, int n, int dt,
                                                int64_t K, int64_t ldx, void* ws, llmc_stream_t stream) {
    const llmc_hessian_problem_t h = one_problem(nullptr, X_list_host, T_list_host, n, K, ldx, 0.0, 1.0);
    return llmc_hessian_accum_multi_partials(&h, 1, dt, ws, stream);
}

extern "C" int llmc_hessian_accum_barrier_timeouts(const void* ws, const int64_t* T_list_host, int n, int64_t K, int64_t ldx,
                                                   unsigned* out_host, llmc_stream_t stream) {
    const llmc_hessian_problem_t h = one_problem(nullptr, nullptr, T_list_host, n, K, ldx, 0.0, 1.0);
    return llmc_hessian_accum_multi_barrier_timeouts(&h, 1, ws, out_host, stream);
}

extern "C" int llmc_hessian_accum_ptrs_reduce(float* H, const int64_t* T_list_host, int n, int64_t K, int64_t ldx,
                                              double n_before, double n_after, const void* ws, llmc_stream_t stream) {
    const llmc_hessian_problem_t h = one_problem(H, nullptr, T_list_host, n, K, ldx, n_before, n_after);
    return llmc_hessian_accum_multi_reduce(&h, 1, ws, stream);
}

extern "C" int llmc_hessian_accum_ptrs(float* H, const void* const* X_list_host, const int64_t* T_list_host, int n, int dt,
                                       int64_t K, int64_t ldx, double n_before, double n_after, void* ws,
                                       llmc_stream_t stream) {
    LLMC_REQUIRE(H != nullptr, "hessian_accum: null H");
    int rc = llmc_hessian_accum_ptrs_partials(X_list_host, T_list_host, n, dt, K, ldx, ws, stream);
    if (rc) return rc;
    return llmc_hessian_accum_ptrs_reduce(H, T_list_host, n, K, ldx, n_before, n_after, ws, stream);
}

extern "C" int llmc_hessian_accum_partials(const void* X, int dt, int64_t T, int64_t K, int64_t ldx, void* ws,
                                           llmc_stream_t stream) {
    return llmc_hessian_accum_ptrs_partials(&X, &T, 1, dt, K, ldx, ws, stream);
}

extern "C" int llmc_hessian_accum_reduce(float* H, int64_t T, int64_t K, int64_t ldx, double n_before,
                                         double n_after, const void* ws, llmc_stream_t stream) {
    return llmc_hessian_accum_ptrs_reduce(H, &T, 1, K, ldx, n_before, n_after, ws, stream);
}

extern "C" int llmc_hessian_accum(float* H, const void* X, int dt, int64_t T, int64_t K, int64_t ldx,
                                  double n_before, double n_after, void* ws, llmc_stream_t stream) {
    return llmc_hessian_accum_ptrs(H, &X, &T, 1, dt, K, ldx, n_before, n_after, ws, stream);
}
