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
static constexpr int SYRK_MAX_SAMPLES = 192;     // table entries per launch (the whole argument block stays < 4 KiB)
static constexpr int SYRK_MAX_CHUNKS = 32;
static constexpr int SYRK_UNIT_SAMPLES = 64;     // samples one unit may cross (one per lane; the kernel clamps at 63)

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

struct SyrkArgs {
    int64_t ldx;       // row stride of every sample, elements
    int K;
    int nb;            // ceil(K / 256)
    int ntiles_p;      // padded tile count (tiles_padded(nb))
    int S;             // token chunks
    int n;             // samples
    int pad_;
    float* part;       // [S * ntiles_p][256*256] fp32, fragment order
    unsigned* sync;    // round barrier counter (zeroed before the launch), or null
    uint32_t cb[SYRK_MAX_CHUNKS + 1];   // chunk s = groups [cb[s], cb[s + 1]) of the padded token axis (even boundaries)
    uint32_t ci[SYRK_MAX_CHUNKS + 1];   // the sample that holds group cb[s]
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
    const int64_t row_bytes = a.ldx * 2;
    const int64_t group_bytes = (int64_t)GROUP_TOK * row_bytes;

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
    const int nunits = a.S * a.ntiles_p;
    const int nrounds = (nunits + G - 1) / G;
    const uint32_t slab = (uint32_t)(8 * row_bytes);
    const uint32_t stage_bytes = (uint32_t)(S4_TOK * row_bytes);
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
        const int s = u / a.ntiles_p;
        const int ti = u - s * a.ntiles_p;
        const TileIdx t = decode_tile(ti, a.nb);
        if (!t.valid) continue;
        const uint32_t gb = a.cb[s];
        const int ngroups = (int)(a.cb[s + 1] - gb);
        const int i0 = (int)a.ci[s];

        // lane table: descriptor words of sample i0 + lane as this unit sees it (file header, "sample table")
        int v0, v1, v2, vend;
        {
            int li = i0 + lane;
            const bool ok = li < a.n;
            if (!ok) li = a.n - 1;
            const SyrkSample e = a.smp[li];
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
        const uint32_t vB = (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bj * TM + ch_off) * 2);

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
                    else if constexpr (f <= 4) fb[f - 1] = tr_frag(lds + offB[SL >> 1][f - 1], IMM + kk * 16 * TM * 2);
                    else fa[f - 4] = tr_frag(lds + offA[SL >> 1][f - 4], IMM + kk * 16 * TM * 2);
                }
            };
            // one burst: 16 MFMAs on (fa, fb). Behind MFMA i: i in {0,1,2,4,5,6,8,9} -> the next fragment of slice kk
            // of slot RSL into (na, nb) (all eight are back six MFMAs before the burst ends); i in {3,7,11,15} ->
            // LDS-DMA piece D0 + i/4 of slot DSL (one KiB per wave every four MFMAs = 32 B/clk per CU, evenly spread);
            // i in {10,12,13,14} -> `extra(i)`: scalar bookkeeping that must not cost an issue slot of its own
            auto burst = [&](const s16x8 (&fa)[4], const s16x8 (&fb)[4], s16x8 (&na)[4], s16x8 (&nb)[4], auto rslc, int rd_kk,
                             auto dslc, auto d0c, const i32x4& rs, auto&& extra) {
                constexpr int D0 = decltype(d0c)::value;
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // serpentine walk of the 4x4 accumulator block: one operand changes per step
                    constexpr int mi = i >> 2, ni = (mi & 1) ? 3 - (i & 3) : (i & 3);
                    acc[mi][ni] = Mfma<DT>::run(fa[mi], fb[ni], acc[mi][ni]);
                    if constexpr ((i & 3) == 3) piece(dslc, std::integral_constant<int, D0 + (i >> 2)>{}, rs);
                    else if constexpr (i < 10) frag(rslc, rd_kk, std::integral_constant<int, i - (i >> 2)>{}, na, nb);
                    else extra(ic);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            auto nothing = [](auto) {};
            // one group (ring turn) g: `cur` = descriptor of group g, `nxt` = of group g + 1; `cur` is rebuilt for group
            // g + 2 behind the second burst of the group's first stage
            auto group = [&](i32x4& cur, const i32x4& nxt, int g) {
                static_for<0, NSLOT>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;              // stage st = g*NSLOT + J sits in slot J
                    constexpr int JN = (J + 1) % NSLOT;                 // slot of stage st+1
                    constexpr int JP = (J + NSLOT - 1) % NSLOT;         // slot of stage st+NSLOT-1 (= st-1)
                    // slice 0 of stage st; fetches slice 1; requests the B half of stage st+NSLOT-1 (J = 0: the last
                    // stage of this group, otherwise a stage of the next group)
                    burst(fa0, fb0, fa1, fb1, jc, 1, std::integral_constant<int, JP>{}, std::integral_constant<int, 4>{},
                          J == 0 ? cur : nxt, nothing);
                    // this wave's pieces of stage st+1 have landed (NSLOT-2 later stages may still be in flight);
                    // every wave has read the whole of stage st once its lgkmcnt(0) is behind the barrier
                    dma_wait_upto<(NSLOT - 2) * PER>();
                    lds_wait_all();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // slice 1; fetches slice 0 of stage st+1; requests the A half of stage st+NSLOT (next group) into slot J
                    if constexpr (J == 0) {
                        int rc = 0;
                        burst(fa1, fb1, fa0, fb0, std::integral_constant<int, JN>{}, 0, jc, std::integral_constant<int, 0>{}, nxt,
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
                        burst(fa1, fb1, fa0, fb0, std::integral_constant<int, JN>{}, 0, jc, std::integral_constant<int, 0>{}, nxt,
                              nothing);
                    }
                });
            };
            // prologue: stages 0 .. NSLOT-2 and the A half of stage NSLOT-1 (all of group 0) requested, stage 0 published
            static_for<0, NSLOT - 1>([&](auto slc) { static_for<0, 8>([&](auto dc) { piece(slc, dc, dA); }); });
            static_for<0, 4>([&](auto dc) { piece(std::integral_constant<int, NSLOT - 1>{}, dc, dA); });
            dma_wait_upto<(NSLOT - 2) * PER + PER / 2>();
            __builtin_amdgcn_s_barrier();
            static_for<0, 8>([&](auto fc) { frag(std::integral_constant<int, 0>{}, 0, fc, fa0, fb0); });
            // groups in pairs (the descriptor sets swap roles); an odd chunk gets one padding group of zeros
            for (int g = 0; g < ngroups; g += 2) {
                group(dA, dB, g);
                group(dB, dA, g + 1);
            }
            lds_wait_all();                  // the trailing fragment reads
            __builtin_amdgcn_s_barrier();    // ... of every wave, before the next unit's prologue overwrites the ring
        }

        // partial tile in fragment order of a 2x4 wave grid with 4x2 accumulators per wave (what k_syrk_fixup decodes):
        // wave (wm, wn) x accumulator (m, n) of the 2x2 / 4x4 layout is wave (wm, 2*wn + n/2) x accumulator (m, n%2)
        float* slot = a.part + (int64_t)u * TILE_FLOATS;
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
        // the stores share the VM counter with the next unit's LDS-DMA: drain them (and the trailing requests)
        dma_wait_all();
    }
}

// Sum the S partials of each tile (chunk order), H <- alpha*H + beta*sum, mirror to the upper triangle.
// One thread per float4 of the fragment-order tile: idx -> (wv,m,n,q,lane) -> rows i0..i0+3, column j.
__global__ __launch_bounds__(256) void k_syrk_fixup(const float* __restrict__ part, float* __restrict__ H,
                                                    int K, int nb, int ntiles_p, int S, float alpha,
                                                    float beta) {
    const int ti = blockIdx.y;
    const TileIdx t = decode_tile(ti, nb);
    if (!t.valid) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;  // 0 .. 16383
    const int lane = idx & 63;
    const int q = (idx >> 6) & 3;
    const int n = (idx >> 8) & 1;
    const int m = (idx >> 9) & 3;
    const int wv = idx >> 11;
    const int wm = wv >> 2, wn = wv & 3;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float* slot = part + ((int64_t)s * ntiles_p + ti) * TILE_FLOATS;
        f32x4 v = *reinterpret_cast<const f32x4*>(slot + (int64_t)idx * 4);
        sum += v;
    }
    const int i0 = t.bi * TM + wm * 128 + m * 32 + 8 * q + 4 * (lane >> 5);
    const int j = t.bj * TM + wn * 64 + n * 32 + (lane & 31);
    if (j >= K) return;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = i0 + r;
        if (i < K) {
            float h = beta * sum[r];
            if (alpha != 0.0f) h += alpha * H[(int64_t)i * K + j];
            o[r] = h;
            H[(int64_t)i * K + j] = h;
        }
    }
    if (t.bi != t.bj) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i0 + r < K) H[(int64_t)j * K + i0 + r] = o[r];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Exact diagonal (round 5, opt-in): d[j] = (n_before / n_after) d[j] + (2 / n_after) sum_t X[t][j]^2 with the sum formed in
// fp64 (squares of 16-bit values are exact in fp32; eight of them are added in fp32 — 19 bits of a 24-bit significand unless
// their exponents differ by more than 5 — then folded into an fp64 accumulator), and H[j][j] = (float) d[j].
// k_syrk4 accumulates 16 products per MFMA into fp32 over 30 - 65 k tokens per unit: its diagonal carries 2 - 3e-6 of relative
// noise, twice what the reference's sgemm leaves (profiles/r04_parity_envelope_full_down.txt); diag(H) is what GPTQ's actorder
// sorts and what the damping averages. One more pass over X (HBM-bound: 2 T K bytes), which is why it is opt-in.
// Grid: x = 512-column blocks, y = token slices; a wave reads whole 1-KiB row segments (16 B per lane).
// ---------------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void k_diag_sumsq(const SyrkArgs a, double* __restrict__ part /* [gridDim.y][K] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 512 + lane * 8;
    double acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0;
    if (c0 < a.K) {
        const int nslice = gridDim.y;
        for (int si = 0; si < a.n; ++si) {
            const char* base = (const char*)(uintptr_t)a.smp[si].base;
            const int64_t T = a.smp[si].T;
            // rows of this sample dealt to (slice, wave) round-robin in runs of 8 (one fp32 partial per run)
            for (int64_t r0 = ((int64_t)blockIdx.y * 4 + wv) * 8; r0 < T; r0 += (int64_t)nslice * 32) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = 0.0f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r0 + r < T) {
                        const uint4 v = *reinterpret_cast<const uint4*>(base + ((r0 + r) * a.ldx + c0) * 2);
                        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float x0, x1;
                            if constexpr (DT == LLMC_BF16) {
                                x0 = __uint_as_float(w[q] << 16);
                                x1 = __uint_as_float(w[q] & 0xffff0000u);
                            } else {
                                x0 = f16_bits_to_f32((uint16_t)(w[q] & 0xffffu));
                                x1 = f16_bits_to_f32((uint16_t)(w[q] >> 16));
                            }
                            f[2 * q] = __builtin_fmaf(x0, x0, f[2 * q]);
                            f[2 * q + 1] = __builtin_fmaf(x1, x1, f[2 * q + 1]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (double)f[e];
            }
        }
    }
    // the four waves of the workgroup hold disjoint rows of the same columns: sum them through LDS
    __shared__ double red[4][512];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wv][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int64_t col = (int64_t)blockIdx.x * 512 + c;
        if (col < a.K) part[(int64_t)blockIdx.y * a.K + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

__global__ __launch_bounds__(256) void k_diag_apply(const double* __restrict__ part, int nslice, int K, double alpha, double beta,
                                                    double* __restrict__ dstate, float* __restrict__ H) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= K) return;
    double s = 0.0;
    for (int i = 0; i < nslice; ++i) s += part[(int64_t)i * K + j];
    const double d = alpha * (alpha != 0.0 ? dstate[j] : 0.0) + beta * s;
    dstate[j] = d;
    H[(int64_t)j * K + j] = (float)d;
}

static inline int choose_chunks(int ntiles_real, int64_t ngroups, int64_t x_bytes, int ncu) {
    // pick S >= Smin minimising a simple time model (microseconds):
    //   rounds * (groups per unit * t_group + t_unit) + fixup traffic (S partial tiles written + read)
    const double t_group = 1.8, t_unit = 6.0, fix_us_per_tile = 0.13;  // 2 x 256 KiB at ~4 TB/s
    int smin = (int)(x_bytes / (1ll << 31)) + 1;   // unit-relative byte offsets stay below 2^31
    if (smin > ngroups) smin = (int)ngroups;
    if (smin < 1) smin = 1;
    int best = smin;
    double best_cost = 1e30;
    for (int S = smin; S <= SYRK_MAX_CHUNKS && S <= ngroups; ++S) {
        int64_t units = (int64_t)ntiles_real * S;
        double rounds = (double)ceil_div64(units, ncu);
        double cost = rounds * ((double)ceil_div64(ngroups, S) * t_group + t_unit) + fix_us_per_tile * (double)units;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = S;
        }
    }
    return best;
}

}  // namespace llmc

using namespace llmc;

// Geometry of one launch: the padded token axis, the chunks, and which sample each chunk starts in. Fills the
// geometry fields of `a` (not the pointers). Returns 0, or LLMC_EINVAL with the message set.
static int syrk_plan(const int64_t* T_list, int n, int64_t K, int64_t ldx, SyrkArgs* a) {
    LLMC_REQUIRE(T_list && n >= 1, "hessian_accum: empty sample list");
    LLMC_REQUIRE(n <= SYRK_MAX_SAMPLES, "hessian_accum: more than LLMC_HESSIAN_MAX_SAMPLES samples in one call");
    LLMC_REQUIRE(K > 0 && K < (1 << 30) && ldx >= K && ldx % 8 == 0, "hessian_accum: rows must be 16-B aligned, K < 2^30");
    int64_t G = 0;
    for (int i = 0; i < n; ++i) {
        LLMC_REQUIRE(T_list[i] > 0 && T_list[i] < (1ll << 31), "hessian_accum: every sample needs 0 < tokens < 2^31");
        a->smp[i].T = (uint32_t)T_list[i];
        a->smp[i].g0 = (uint32_t)G;
        G += ceil_div64(T_list[i], GROUP_TOK);
    }
    LLMC_REQUIRE(G < (1ll << 31), "hessian_accum: too many tokens in one call");
    a->ldx = ldx;
    a->K = (int)K;
    a->nb = (int)ceil_div64(K, TM);
    a->ntiles_p = tiles_padded(a->nb);
    a->n = n;
    a->pad_ = 0;
    const int real = a->nb * (a->nb + 1) / 2;
    int S = choose_chunks(real, G, G * GROUP_TOK * ldx * 2, 256);
#ifdef LLMC_LAB
    if (const char* e = getenv("LLMC_SYRK_S")) S = atoi(e);   // lab: force the token-chunk count
#endif
    if (S > G / 2) S = (int)(G / 2);     // every chunk at least one group pair
    if (S < 1) S = 1;
    for (;; ++S) {
        LLMC_REQUIRE(S <= SYRK_MAX_CHUNKS && (S <= G / 2 || S == 1), "hessian_accum: samples too short for one call (stage them into fewer, longer ones)");
        int i = 0;
        bool fits = true;
        for (int s = 0; s <= S; ++s) {
            const int64_t g = s == S ? G : (s * G / S) & ~(int64_t)1;   // interior boundaries even: the kernel walks group pairs
            a->cb[s] = (uint32_t)g;
            while (i + 1 < n && (int64_t)a->smp[i + 1].g0 <= g) ++i;
            a->ci[s] = (uint32_t)i;
            // a unit walks the samples ci[s] .. (the one holding group cb[s + 1], whose first stages it requests)
            if (s > 0 && (int)a->ci[s] - (int)a->ci[s - 1] + 1 > SYRK_UNIT_SAMPLES) fits = false;
        }
        if (fits) break;
    }
    a->S = S;
    return LLMC_OK;
}

static size_t syrk_ws_bytes(const SyrkArgs& a) {
    return (size_t)a.S * a.ntiles_p * TILE_FLOATS * sizeof(float) + 256;
}

extern "C" int llmc_hessian_max_samples(void) { return SYRK_MAX_SAMPLES; }

extern "C" size_t llmc_hessian_accum_ptrs_ws_bytes(const int64_t* T_list_host, int n, int64_t K, int64_t ldx) {
    SyrkArgs a;
    if (syrk_plan(T_list_host, n, K, ldx, &a)) return 0;
    return syrk_ws_bytes(a);
}

extern "C" size_t llmc_hessian_accum_ws_bytes(int64_t T, int64_t K, int64_t ldx) {
    return llmc_hessian_accum_ptrs_ws_bytes(&T, 1, K, ldx);
}

extern "C" int llmc_hessian_accum_ptrs_partials(const void* const* X_list_host, const int64_t* T_list_host, int n, int dt,
                                                int64_t K, int64_t ldx, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "hessian_accum: X must be f16 or bf16");
    LLMC_REQUIRE(X_list_host && ws, "hessian_accum: null argument");
    SyrkArgs a;
    int rc = syrk_plan(T_list_host, n, K, ldx, &a);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        LLMC_REQUIRE(X_list_host[i] && ((uintptr_t)X_list_host[i] & 15) == 0, "hessian_accum: X rows must be 16-B aligned");
        a.smp[i].base = (uint64_t)(uintptr_t)X_list_host[i];
    }
    hipStream_t st = (hipStream_t)stream;
    a.part = (float*)ws;
    a.sync = (unsigned*)((char*)ws + (size_t)a.S * a.ntiles_p * TILE_FLOATS * sizeof(float));
    int abl = 0;
#ifdef LLMC_LAB
    if (getenv("LLMC_SYRK_NOSYNC")) a.sync = nullptr;
    if (const char* e = getenv("LLMC_SYRK_ABL")) abl = atoi(e);   // wrong results by design
#endif
    if (a.sync) LLMC_HIP_CHECK(hipMemsetAsync(a.sync, 0, 8, st));     // round counter + time-out counter
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
    rc = ensure_dynamic_lds(fn, lds_bytes);
    if (rc) return rc;
    void* kargs[] = {(void*)&a};
    LLMC_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(S4_THREADS), kargs, (size_t)lds_bytes, st));
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

// Number of round barriers of the LAST launch on this workspace that gave up waiting (device word behind the partials):
// copies it to *out_host after synchronising `stream`. 0 in a healthy run; > 0 means another stream kept workgroups of the
// persistent grid off their CUs (results stay correct, the kernel re-fetches its panels: slower).
extern "C" int llmc_hessian_accum_barrier_timeouts(const void* ws, const int64_t* T_list_host, int n, int64_t K, int64_t ldx,
                                                   unsigned* out_host, llmc_stream_t stream) {
    LLMC_REQUIRE(ws && out_host, "hessian_accum_barrier_timeouts: null argument");
    SyrkArgs a;
    int rc = syrk_plan(T_list_host, n, K, ldx, &a);
    if (rc) return rc;
    const char* p = (const char*)ws + (size_t)a.S * a.ntiles_p * TILE_FLOATS * sizeof(float) + 4;
    LLMC_HIP_CHECK(hipMemcpyAsync(out_host, p, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    LLMC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return LLMC_OK;
}

static constexpr int DIAG_SLICES = 96;       // token slices: 96 x ceil(K / 512) workgroups

extern "C" size_t llmc_hessian_diag_ws_bytes(int64_t K) {
    return K > 0 ? (size_t)DIAG_SLICES * K * sizeof(double) : 0;
}

// Optional second pass of an accumulation step (see k_diag_sumsq): dstate [K] fp64 carries the exact running diagonal across
// calls (ignored and overwritten when n_before == 0); H's diagonal is overwritten with its fp32 rounding. Call it AFTER the
// llmc_hessian_accum* call of the same samples, with the same n_before / n_after. ws: llmc_hessian_diag_ws_bytes(K).
extern "C" int llmc_hessian_diag_accum_ptrs(float* H, double* dstate, const void* const* X_list_host, const int64_t* T_list_host,
                                            int n, int dt, int64_t K, int64_t ldx, double n_before, double n_after, void* ws,
                                            llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "hessian_diag: X must be f16 or bf16");
    LLMC_REQUIRE(H && dstate && X_list_host && T_list_host && ws && n_after > 0, "hessian_diag: null argument");
    LLMC_REQUIRE(n >= 1 && n <= SYRK_MAX_SAMPLES, "hessian_diag: 1 .. LLMC_HESSIAN_MAX_SAMPLES samples per call");
    LLMC_REQUIRE(K > 0 && K % 8 == 0 && ldx >= K && ldx % 8 == 0, "hessian_diag: K and the row stride must be multiples of 8");
    SyrkArgs a;
    a.K = (int)K; a.ldx = ldx; a.n = n;
    for (int i = 0; i < n; ++i) {
        LLMC_REQUIRE(X_list_host[i] && ((uintptr_t)X_list_host[i] & 15) == 0 && T_list_host[i] > 0, "hessian_diag: bad sample");
        a.smp[i].base = (uint64_t)(uintptr_t)X_list_host[i];
        a.smp[i].T = (uint32_t)T_list_host[i];
        a.smp[i].g0 = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)ceil_div64(K, 512), DIAG_SLICES);
    if (dt == LLMC_BF16) hipLaunchKernelGGL((k_diag_sumsq<LLMC_BF16>), grid, dim3(256), 0, st, a, (double*)ws);
    else hipLaunchKernelGGL((k_diag_sumsq<LLMC_F16>), grid, dim3(256), 0, st, a, (double*)ws);
    LLMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_diag_apply, dim3((unsigned)ceil_div64(K, 256)), dim3(256), 0, st, (const double*)ws, DIAG_SLICES, (int)K,
                       n_before / n_after, 2.0 / n_after, dstate, H);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_hessian_accum_ptrs_reduce(float* H, const int64_t* T_list_host, int n, int64_t K, int64_t ldx,
                                              double n_before, double n_after, const void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(H && ws && n_after > 0, "hessian_accum_reduce: bad argument");
    SyrkArgs a;
    int rc = syrk_plan(T_list_host, n, K, ldx, &a);
    if (rc) return rc;
    float alpha = (float)(n_before / n_after);
    float beta = (float)(2.0 / n_after);
    hipLaunchKernelGGL(k_syrk_fixup, dim3(TILE_FLOATS / 4 / 256, a.ntiles_p), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, H, (int)K, a.nb, a.ntiles_p, a.S, alpha, beta);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
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
