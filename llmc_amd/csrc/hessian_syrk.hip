// hessian_syrk.hip — K1: H <- a*H + b*X^T X on the MFMA pipe (GPTQ.add_batch, gptq.py:254-295).
//
// X is [T tokens, K channels] 16-bit, channel-contiguous, so BOTH MFMA operands are strided along the
// reduction (token) axis. Tiles are therefore staged token-major in LDS exactly as they lie in HBM
// (LDS-DMA, 16 B per lane) and transposed on the way to the registers by ds_read_b64_tr_b16.
//
// Decomposition
//   output tile   256 x 256 channels (lower triangle of the tile grid only), 8 waves as 2(M) x 4(N),
//                 each wave 128 x 64 = 4 x 2 MFMA 32x32x16 accumulators (128 VGPRs)
//   K-step        64 tokens: A panel [64][256] + B panel [64][256] 16-bit = 64 KiB, 2 stages = 128 KiB LDS
//   unit          (tile, token chunk s of S); units are dealt round-robin to a persistent grid of one
//                 workgroup per CU so that the 32 workgroups of an XCD sit on 32 consecutive tiles of a
//                 4x4-superblock order (12 shared panels in the XCD's L2) at the SAME token position.
//   partials      every unit writes its 256x256 fp32 partial in MFMA-fragment order (16-B stores) to the
//                 workspace; k_syrk_fixup sums the S partials of a tile in chunk order (deterministic),
//                 applies H <- a*H + b*sum and mirrors the tile to the upper triangle.
// LDS bank layout: a 4-token x 64-B tr16 read by a 32-lane half hits 4 rows at a 512-B stride; the 64-B
// unit index is XOR-ed with (token & 3) so the four rows land on four different bank quarters. The
// swizzle is applied to the DMA's per-lane SOURCE address (LDS image stays lane-linear) and to the read.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "mfma_common.h"

namespace llmc {

static constexpr int TM = 256;       // tile edge (channels)
static constexpr int BK = 64;        // tokens per K-step
static constexpr int PANEL_BYTES = BK * TM * 2;  // 32 KiB
static constexpr int STAGE_BYTES = 2 * PANEL_BYTES;
static constexpr int SYRK_THREADS = 512;
static constexpr int SYRK_LDS = 2 * STAGE_BYTES;  // 128 KiB
static constexpr int TILE_FLOATS = TM * TM;

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

struct SyrkArgs {
    const char* X;     // [T, K] 16-bit, row stride ldx elements
    int64_t T;
    int64_t ldx;       // elements
    int K;
    int nb;            // ceil(K / 256)
    int ntiles_p;      // padded tile count (tiles_padded(nb))
    int S;             // token chunks
    int nk;            // ceil(T / 64)
    float* part;       // [S * ntiles_p][256*256] fp32, fragment order
    unsigned* sync;    // round barrier counter (zeroed before the launch), or null
    int kalign;        // chunk boundaries are multiples of this many K-steps (k_syrk4 works in whole ring groups)
};

// First 64-token K-step of chunk s (s = S gives the end of the last one). Interior boundaries are rounded down to a
// multiple of kalign, the end is rounded up: K-steps past ceil(T/64) read rows past T, which the buffer descriptor
// zero-fills, so they add nothing.
__host__ __device__ inline int chunk_begin(int s, int nk, int S, int kalign) {
    if (s >= S) return (nk + kalign - 1) / kalign * kalign;
    return (int)(((int64_t)s * nk) / S) / kalign * kalign;
}

template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PH8 = false: one barrier per K-step, the whole next stage is requested at its start and waited for (vmcnt 0) at
// its end. PH8 = true: the guide's phase-split schedule (cdna_hip_programming.md, "8-phase template") mapped onto this
// kernel's 64-token K-step: four phases per K-step, one per 16-token slice,
//     phase p:  request slice p of the NEXT K-step (2 LDS-DMA pieces) | read the 6 fragments of slice p |
//               counted vmcnt: retire slice p+1 | s_barrier | lgkmcnt(0) | setprio(1) 8 MFMA setprio(0) | s_barrier
// and the two wave rows (wm = 0 / 1, one wave of each per SIMD) run offset by one barrier, so that one of them is in
// its MFMA burst while the other issues its reads and requests: the matrix pipe is fed alternately and never waits
// for a whole stage. A slice is requested a full K-step before it is read and retired (own vmcnt, then a barrier)
// one phase before it is read; it overwrites LDS last read a full K-step earlier.
template <int DT, bool PH8>
__global__ __launch_bounds__(SYRK_THREADS) void k_syrk(SyrkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;

    // ---- DMA geometry: instruction q of wave wv fills LDS KiB-block (q*8 + wv) of a stage: blocks 0..31
    // are the A panel (2 token rows each), 32..63 the B panel. Lane l -> row (l>>5), physical 16-B chunk (l&31).
    const int lr = lane >> 5;
    const int c16 = lane & 31;
    const int row_lo = 2 * wv + lr;  // token row inside the 16-row slab this instruction group covers
    const int u_log = (c16 >> 2) ^ (row_lo & 3);        // logical 64-B unit held by this physical slot
    const int ch_off = (u_log * 4 + (c16 & 3)) * 8;      // logical channel offset inside the panel
    const int64_t row_bytes = a.ldx * 2;

    // ---- fragment read geometry (see file header and tools/probes/probe_mfma_tr16.hip, H1/H2)
    const int p = lane & 15;
    const int trow = 8 * (lane >> 5) + (p >> 2);              // + 16*kk (+4 for the second read)
    const int sub = 32 * ((lane >> 4) & 1) + 8 * (p & 3);     // byte offset inside the 64-B unit
    int offA[4], offB[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) offA[m] = trow * (TM * 2) + (((4 * wm + m) ^ (p >> 2)) << 6) + sub;
#pragma unroll
    for (int n = 0; n < 2; ++n)
        offB[n] = trow * (TM * 2) + (((2 * wn + n) ^ (p >> 2)) << 6) + sub;

    const int G = gridDim.x;
    const int lw = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // XCD-contiguous logical id
    const int nunits = a.S * a.ntiles_p;

    const int nrounds = (nunits + G - 1) / G;
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
            }
            __syncthreads();
        }
        const int u = lw + round * G;
        if (u >= nunits) continue;
        const int s = u / a.ntiles_p;
        const int ti = u - s * a.ntiles_p;
        const TileIdx t = decode_tile(ti, a.nb);
        if (!t.valid) continue;
        const bool diag = t.bi == t.bj;
        const int ks0 = chunk_begin(s, a.nk, a.S, a.kalign);
        const int ks1 = chunk_begin(s + 1, a.nk, a.S, a.kalign);

        // buffer descriptor over the chunk's rows: reads past row T return 0 (token tail), channel
        // overrun past K only pollutes outputs that the fixup never stores.
        const char* base = a.X + (int64_t)ks0 * BK * row_bytes;
        int64_t rem_bytes = (a.T - (int64_t)ks0 * BK) * row_bytes;
        if (rem_bytes < 0) rem_bytes = 0;
        const uint32_t nrec = rem_bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)rem_bytes;
        i32x4 rsrc;
        rsrc[0] = (int)(uint32_t)(uintptr_t)base;
        rsrc[1] = (int)((uint32_t)((uintptr_t)base >> 32) & 0xffffu);  // stride 0
        rsrc[2] = (int)nrec;
        rsrc[3] = 0x00020000;
        const uint32_t vA = (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bi * TM + ch_off) * 2);
        const uint32_t vB = (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bj * TM + ch_off) * 2);
        const uint32_t slab = (uint32_t)(16 * row_bytes);  // 16 token rows per instruction index q

        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        auto stage = [&](int buf, int ks) {
            const uint32_t koff = (uint32_t)((int64_t)(ks - ks0) * BK * row_bytes);
            const uint32_t dst = lds_base + buf * STAGE_BYTES + wv * 1024;
#pragma unroll
            for (int q = 0; q < 4; ++q) dma16(rsrc, vA + koff + q * slab, dst + q * 8192);
            if (!diag) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dma16(rsrc, vB + koff + q * slab, dst + PANEL_BYTES + q * 8192);
            }
        };

        if constexpr (PH8) {
            const int nks = ks1 - ks0;
            if (nks > 0) {
                stage(0, ks0);
                dma_wait_all();
                __builtin_amdgcn_s_barrier();
                if (wm == 1) __builtin_amdgcn_s_barrier();   // offset the second wave row by one barrier
                for (int i = 0; i < nks; ++i) {
                    const bool more = i + 1 < nks;
                    LDS_AS char* pa = lds + (i & 1) * STAGE_BYTES;
                    LDS_AS char* pb = diag ? pa : pa + PANEL_BYTES;
                    const uint32_t koff = (uint32_t)((int64_t)(i + 1) * BK * row_bytes);
                    const uint32_t dst = lds_base + ((i + 1) & 1) * STAGE_BYTES + wv * 1024;
                    auto phase = [&](auto phc) {
                        constexpr int ph = decltype(phc)::value;
                        if (more) {
                            dma16(rsrc, vA + koff + ph * slab, dst + ph * 8192);
                            if (!diag) dma16(rsrc, vB + koff + ph * slab, dst + PANEL_BYTES + ph * 8192);
                        }
                        s16x8 fa[4], fb[2];
#pragma unroll
                        for (int m = 0; m < 4; ++m) fa[m] = tr_frag(pa + offA[m], ph * 16 * TM * 2);
#pragma unroll
                        for (int n = 0; n < 2; ++n) fb[n] = tr_frag(pb + offB[n], ph * 16 * TM * 2);
                        // retire slice ph+1 (of this K-step, or slice 0 of the next one): requests issued after it
                        // may stay in flight — three phases' worth while requests are being issued, fewer at the end
                        if (more) {
                            if (diag) vm_wait<3>(); else vm_wait<6>();
                        } else if (ph < 3) {
                            if (diag) vm_wait<2 - ph>(); else vm_wait<2 * (2 - ph)>();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int n = 0; n < 2; ++n) acc[m][n] = Mfma<DT>::run(fa[m], fb[n], acc[m][n]);
                        __builtin_amdgcn_s_setprio(0);
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                    };
                    phase(std::integral_constant<int, 0>{});
                    phase(std::integral_constant<int, 1>{});
                    phase(std::integral_constant<int, 2>{});
                    phase(std::integral_constant<int, 3>{});
                }
                if (wm == 0) __builtin_amdgcn_s_barrier();   // the first wave row waits for the second one's last phase
            }
        }
        int cur = 0;
        if (!PH8 && ks0 < ks1) stage(0, ks0);
        for (int ks = ks0; !PH8 && ks < ks1; ++ks) {
            dma_wait_all();   // this wave's DMA pieces have landed
            __syncthreads();  // everyone's pieces landed; previous stage fully read
            if (ks + 1 < ks1) stage(cur ^ 1, ks + 1);
            LDS_AS char* pa = lds + cur * STAGE_BYTES;
            LDS_AS char* pb = diag ? pa : pa + PANEL_BYTES;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                s16x8 fa[4], fb[2];
#pragma unroll
                for (int m = 0; m < 4; ++m) fa[m] = tr_frag(pa + offA[m], kk * 16 * TM * 2);
#pragma unroll
                for (int n = 0; n < 2; ++n) fb[n] = tr_frag(pb + offB[n], kk * 16 * TM * 2);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = Mfma<DT>::run(fa[m], fb[n], acc[m][n]);
            }
            cur ^= 1;
        }
        __syncthreads();  // all waves done with LDS before the next unit's first stage

        // ---- partial tile in fragment order: ((wv*4+m)*2+n)*4+q -> 64 lanes x float4
        float* slot = a.part + (int64_t)u * TILE_FLOATS;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2],
                               acc[m][n][4 * q + 3]};
                    int idx = ((((wv * 4 + m) * 2 + n) * 4 + q) * 64 + lane);
                    *reinterpret_cast<f32x4*>(slot + (int64_t)idx * 4) = v;
                }
    }
}

// -----------------------------------------------------------------------------------------------------
// k_syrk2 — same tiling and data path as k_syrk, deeper pipeline:
//   * LDS ring of 4 stages of 32 tokens (A 16 KiB + B 16 KiB each); up to 3 stages of LDS-DMA in flight, waited
//     with a COUNTED vmcnt (never 0 in the steady state);
//   * fragments are double-buffered in registers at 16-token granularity and the barrier that publishes the next
//     stage sits BETWEEN the two MFMA bursts of a stage, so the first fragments of the next stage are fetched
//     under the second burst: the matrix pipe no longer drains at every K-step boundary
//     (k_syrk: barrier -> 8 DMA issues -> first ds_reads -> MFMA, ~25-30 % of the step with the pipe idle).
// Steady state of stage s (slot s & 3), F0 = fragments of tokens 0..15 of the stage, F1 = tokens 16..31:
//     read F1(s) | MFMA F0 | wait DMA(s+1) landed, lgkmcnt(0), s_barrier | issue DMA(s+4) into slot s |
//     read F0(s+1) | MFMA F1
// After the barrier of stage s every wave has completed its reads of stage s (F0 before the previous barrier, F1
// waited by lgkmcnt(0)), so slot s is free for stage s+4.
// -----------------------------------------------------------------------------------------------------
static constexpr int ST_TOK = 32;
static constexpr int ST_PANEL = ST_TOK * TM * 2;     // 16 KiB
static constexpr int ST_BYTES = 2 * ST_PANEL;        // 32 KiB
static constexpr int ST_RING = 4;
static constexpr int SYRK2_LDS = ST_RING * ST_BYTES; // 128 KiB

template <int N> __device__ __forceinline__ void dma_wait_upto() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one unit (tile x token chunk) of k_syrk2; DIAG: the tile is on the diagonal (one panel, 2 DMA pieces per stage)
template <int DT, bool DIAG>
__device__ __forceinline__ void syrk2_unit(f32x16 (&acc)[4][2], LDS_AS char* lds, uint32_t lds_base, i32x4 rsrc,
                                           uint32_t vA, uint32_t vB, uint32_t slab, int64_t row_bytes, int nst,
                                           int wv, const int (&offA)[4], const int (&offB)[2]) {
    constexpr int PER = DIAG ? 2 : 4;   // DMA instructions per stage per wave
    auto stage = [&](int st) {
        const uint32_t koff = (uint32_t)((int64_t)st * ST_TOK * row_bytes);
        const uint32_t dst = lds_base + (st & (ST_RING - 1)) * ST_BYTES + wv * 1024;
        dma16(rsrc, vA + koff, dst);
        dma16(rsrc, vA + koff + slab, dst + 8192);
        if (!DIAG) {
            dma16(rsrc, vB + koff, dst + ST_PANEL);
            dma16(rsrc, vB + koff + slab, dst + ST_PANEL + 8192);
        }
    };
    s16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    auto read_frags = [&](int st, int kk, s16x8 (&fa)[4], s16x8 (&fb)[2]) {
        LDS_AS char* pa = lds + (st & (ST_RING - 1)) * ST_BYTES;
        LDS_AS char* pb = DIAG ? pa : pa + ST_PANEL;
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = tr_frag(pa + offA[m], kk * 16 * TM * 2);
#pragma unroll
        for (int n = 0; n < 2; ++n) fb[n] = tr_frag(pb + offB[n], kk * 16 * TM * 2);
    };
    auto mma = [&](const s16x8 (&fa)[4], const s16x8 (&fb)[2]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = Mfma<DT>::run(fa[m], fb[n], acc[m][n]);
    };
    if (nst <= 0) return;
    // prologue: fill the ring, publish stage 0, fetch its first fragments
    const int npre = nst < ST_RING ? nst : ST_RING;
    for (int st = 0; st < npre; ++st) stage(st);
    if (npre == ST_RING) dma_wait_upto<3 * PER>(); else dma_wait_upto<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, fa0, fb0);
    // vmcnt counts this wave's DMA instructions in issue order. When stage st+1 is published, the stages issued
    // after it are st+2 and st+3 (steady state): wait until at most 2*PER instructions remain. The last three
    // stages drain with vmcnt(0).
    for (int st = 0; st < nst; ++st) {
        read_frags(st, 1, fa1, fb1);
        mma(fa0, fb0);
        if (st + 3 < nst) dma_wait_upto<2 * PER>(); else dma_wait_upto<0>();
        lds_wait_all();
        __builtin_amdgcn_s_barrier();
        if (st + ST_RING < nst) stage(st + ST_RING);
        if (st + 1 < nst) read_frags(st + 1, 0, fa0, fb0);
        mma(fa1, fb1);
    }
}

template <int DT>
__global__ __launch_bounds__(SYRK_THREADS) void k_syrk2(SyrkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;

    // DMA: instruction q (0,1) of wave wv fills KiB-block (q*8 + wv) of a 16-KiB panel = token rows 2*blk, 2*blk+1
    const int lr = lane >> 5;
    const int c16 = lane & 31;
    const int row_lo = 2 * wv + lr;
    const int u_log = (c16 >> 2) ^ (row_lo & 3);
    const int ch_off = (u_log * 4 + (c16 & 3)) * 8;
    const int64_t row_bytes = a.ldx * 2;

    const int p = lane & 15;
    const int trow = 8 * (lane >> 5) + (p >> 2);
    const int sub = 32 * ((lane >> 4) & 1) + 8 * (p & 3);
    int offA[4], offB[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) offA[m] = trow * (TM * 2) + (((4 * wm + m) ^ (p >> 2)) << 6) + sub;
#pragma unroll
    for (int n = 0; n < 2; ++n) offB[n] = trow * (TM * 2) + (((2 * wn + n) ^ (p >> 2)) << 6) + sub;

    const int G = gridDim.x;
    const int lw = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int nunits = a.S * a.ntiles_p;
    const int nrounds = (nunits + G - 1) / G;
    const int nst_total = (int)((a.T + ST_TOK - 1) / ST_TOK);

    for (int round = 0; round < nrounds; ++round) {
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
            }
            __syncthreads();
        }
        const int u = lw + round * G;
        if (u >= nunits) continue;
        const int s = u / a.ntiles_p;
        const int ti = u - s * a.ntiles_p;
        const TileIdx t = decode_tile(ti, a.nb);
        if (!t.valid) continue;
        // chunk boundaries in 64-token K-steps (same split as k_syrk: ws layout and fixup are shared)
        const int ks0 = chunk_begin(s, a.nk, a.S, a.kalign);
        const int ks1 = chunk_begin(s + 1, a.nk, a.S, a.kalign);
        const int st0 = 2 * ks0;
        int st1 = 2 * ks1;
        if (st1 > nst_total) st1 = nst_total;
        const int nst = st1 - st0;

        const char* base = a.X + (int64_t)st0 * ST_TOK * row_bytes;
        int64_t rem_bytes = (a.T - (int64_t)st0 * ST_TOK) * row_bytes;
        if (rem_bytes < 0) rem_bytes = 0;
        const uint32_t nrec = rem_bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)rem_bytes;
        i32x4 rsrc;
        rsrc[0] = (int)(uint32_t)(uintptr_t)base;
        rsrc[1] = (int)((uint32_t)((uintptr_t)base >> 32) & 0xffffu);
        rsrc[2] = (int)nrec;
        rsrc[3] = 0x00020000;
        const uint32_t vA = (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bi * TM + ch_off) * 2);
        const uint32_t vB = (uint32_t)((int64_t)row_lo * row_bytes + ((int64_t)t.bj * TM + ch_off) * 2);
        const uint32_t slab = (uint32_t)(16 * row_bytes);

        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        if (t.bi == t.bj)
            syrk2_unit<DT, true>(acc, lds, lds_base, rsrc, vA, vB, slab, row_bytes, nst, wv, offA, offB);
        else
            syrk2_unit<DT, false>(acc, lds, lds_base, rsrc, vA, vB, slab, row_bytes, nst, wv, offA, offB);
        dma_wait_all();
        __syncthreads();

        float* slot = a.part + (int64_t)u * TILE_FLOATS;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2],
                               acc[m][n][4 * q + 3]};
                    int idx = ((((wv * 4 + m) * 2 + n) * 4 + q) * 64 + lane);
                    *reinterpret_cast<f32x4*>(slot + (int64_t)idx * 4) = v;
                }
    }
}

// -----------------------------------------------------------------------------------------------------
// k_syrk4 — one wave per SIMD: 4 waves as 2(M) x 2(N), each wave 128 x 128 = 4 x 4 MFMA 32x32x16 accumulators
// (256 accumulator registers; the kernel uses the whole 512-entry file of its SIMD).
//   * 8 fragment reads feed 16 MFMAs per 16-token slice (0.5 per MFMA; the 8-wave kernel's 128x64 wave tile
//     needs 0.75), so the LDS moves a third fewer bytes per flop;
//   * LDS ring of NSLOT stages of 32 tokens (4 -> 128 KiB, 5 -> all 160 KiB), NSLOT-1 stages of LDS-DMA in flight,
//     counted vmcnt (never 0 in the steady state); fragments double-buffered in registers at 16-token granularity;
//     one s_barrier per stage, placed BETWEEN the stage's two MFMA bursts, so every wave arrives with 16 MFMAs' worth
//     of operands already in registers;
//   * the stage loop is unrolled NSLOT times: ring slots are compile-time, LDS-DMA destinations are immediates
//     (the kernel owns the whole LDS, base 0), the scalar source offset of each piece lives in its own SGPR and is
//     advanced right after use — a piece is  s_mov m0 / s_nop 0 / buffer_load ... lds / s_add  with no hazard padding;
//   * every MFMA is followed by at most one fragment read (2 ds_read_b64_tr_b16) or one LDS-DMA piece, pinned in
//     that order: with a single wave per SIMD nothing else hides their issue.
// Same units, same token order per accumulator and the same fragment-order partial tile as k_syrk: results
// are bit-identical to k_syrk's.
// ABL (lab builds only): 1 = no LDS-DMA, 2 = no fragment reads, 4 = LDS-DMA source pinned to the chunk's first stages
// (L2-resident: separates issue cost from miss latency), 8 = pieces without their buffer_load (SALU only).
// -----------------------------------------------------------------------------------------------------
static constexpr int S4_THREADS = 256;
static constexpr int S4_TOK = 32;
static constexpr int S4_PANEL = S4_TOK * TM * 2;      // 16 KiB
static constexpr int S4_STAGE = 2 * S4_PANEL;         // 32 KiB

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

template <int DT, int NSLOT, int ABL>
__global__ __launch_bounds__(S4_THREADS) void k_syrk4(SyrkArgs a) {
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
    const int lw = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int nunits = a.S * a.ntiles_p;
    const int nrounds = (nunits + G - 1) / G;
    const uint32_t slab = (uint32_t)(8 * row_bytes);
    const uint32_t stage_bytes = (uint32_t)(S4_TOK * row_bytes);
    const uint32_t wvoff = (uint32_t)wv * 1024u;

    for (int round = 0; round < nrounds; ++round) {
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
            }
            __syncthreads();
        }
        const int u = lw + round * G;
        if (u >= nunits) continue;
        const int s = u / a.ntiles_p;
        const int ti = u - s * a.ntiles_p;
        const TileIdx t = decode_tile(ti, a.nb);
        if (!t.valid) continue;
        // chunk boundaries in 64-token K-steps, aligned so that a chunk is a whole number of NSLOT-stage groups
        const int ks0 = chunk_begin(s, a.nk, a.S, a.kalign);
        const int ks1 = chunk_begin(s + 1, a.nk, a.S, a.kalign);
        const int st0 = 2 * ks0;
        const int ngroups = (2 * (ks1 - ks0)) / NSLOT;

        const char* base = a.X + (int64_t)st0 * S4_TOK * row_bytes;
        int64_t rem_bytes = (a.T - (int64_t)st0 * S4_TOK) * row_bytes;
        if (rem_bytes < 0) rem_bytes = 0;
        const uint32_t nrec = rem_bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)rem_bytes;
        i32x4 rsrc;
        rsrc[0] = (int)(uint32_t)(uintptr_t)base;
        rsrc[1] = (int)((uint32_t)((uintptr_t)base >> 32) & 0xffffu);
        rsrc[2] = (int)nrec;
        rsrc[3] = 0x00020000;
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
            // scalar source offsets, one per piece: sA[q] belongs to the A half, sB[q] to the B half of a stage
            uint32_t sA[4], sB[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sA[q] = sB[q] = (uint32_t)q * slab;
            // piece d of the stage in ring slot SL: d = 0..3 A-panel KiB-block (d*4 + wv), d = 4..7 the same of B
            auto piece = [&](auto slc, auto dc) {
                constexpr int SL = decltype(slc)::value;
                constexpr int d = decltype(dc)::value;
                if constexpr (DMA) {
                    constexpr int DSTB = SL * S4_STAGE + (d >> 2) * S4_PANEL + (d & 3) * 4 * 1024;
                    if constexpr (d < 4) dma16w<DSTB, LOAD, ADV>(rsrc, vA, sA[d & 3], stage_bytes, wvoff);
                    else dma16w<DSTB, LOAD, ADV>(rsrc, vB, sB[d & 3], stage_bytes, wvoff);
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
            // LDS-DMA piece D0 + i/4 of slot DSL (one KiB per wave every four MFMAs = 32 B/clk per CU, evenly spread)
            auto burst = [&](const s16x8 (&fa)[4], const s16x8 (&fb)[4], s16x8 (&na)[4], s16x8 (&nb)[4], auto rslc, int rd_kk,
                             auto dslc, auto d0c) {
                constexpr int D0 = decltype(d0c)::value;
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // serpentine walk of the 4x4 accumulator block: one operand changes per step
                    constexpr int mi = i >> 2, ni = (mi & 1) ? 3 - (i & 3) : (i & 3);
                    acc[mi][ni] = Mfma<DT>::run(fa[mi], fb[ni], acc[mi][ni]);
                    if constexpr ((i & 3) == 3) piece(dslc, std::integral_constant<int, D0 + (i >> 2)>{});
                    else if constexpr (i < 10) frag(rslc, rd_kk, std::integral_constant<int, i - (i >> 2)>{}, na, nb);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            // prologue: stages 0 .. NSLOT-2 and the A half of stage NSLOT-1 requested, stage 0 published
            static_for<0, NSLOT - 1>([&](auto slc) { static_for<0, 8>([&](auto dc) { piece(slc, dc); }); });
            static_for<0, 4>([&](auto dc) { piece(std::integral_constant<int, NSLOT - 1>{}, dc); });
            dma_wait_upto<(NSLOT - 2) * PER + PER / 2>();
            __builtin_amdgcn_s_barrier();
            static_for<0, 8>([&](auto fc) { frag(std::integral_constant<int, 0>{}, 0, fc, fa0, fb0); });
            for (int g = 0; g < ngroups; ++g) {
                static_for<0, NSLOT>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;              // stage st = g*NSLOT + J sits in slot J
                    constexpr int JN = (J + 1) % NSLOT;                 // slot of stage st+1
                    constexpr int JP = (J + NSLOT - 1) % NSLOT;         // slot of stage st+NSLOT-1 (= st-1)
                    // slice 0 of stage st; fetches slice 1; requests the B half of stage st+NSLOT-1
                    burst(fa0, fb0, fa1, fb1, jc, 1, std::integral_constant<int, JP>{}, std::integral_constant<int, 4>{});
                    // this wave's pieces of stage st+1 have landed (NSLOT-2 later stages may still be in flight);
                    // every wave has read the whole of stage st once its lgkmcnt(0) is behind the barrier
                    dma_wait_upto<(NSLOT - 2) * PER>();
                    lds_wait_all();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // slice 1; fetches slice 0 of stage st+1; requests the A half of stage st+NSLOT into slot J
                    burst(fa1, fb1, fa0, fb0, std::integral_constant<int, JN>{}, 0, jc, std::integral_constant<int, 0>{});
                });
            }
            lds_wait_all();                  // the trailing fragment reads
            __builtin_amdgcn_s_barrier();    // ... of every wave, before the next unit's prologue overwrites the ring
        }

        // partial tile in k_syrk's fragment order: wave (wm, wn) x accumulator (m, n) of the 2x2 / 4x4 layout is
        // wave (wm, 2*wn + n/2) x accumulator (m, n%2) of the 2x4 / 4x2 layout
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

// default kernel and the chunk alignment it needs (a chunk = whole ring groups: 4 stages = 2 K-steps, 5 stages -> 5)
static constexpr int SYRK_DEFAULT_VARIANT = 4;
static constexpr int SYRK_KALIGN = SYRK_DEFAULT_VARIANT == 5 ? 5 : 2;

static inline int choose_chunks(int ntiles_real, int nk, int64_t x_bytes, int ncu) {
    // pick S >= Smin minimising a simple time model (microseconds):
    //   rounds * (K-steps per unit * t_step + t_unit) + fixup traffic (S partial tiles written + read)
    const double t_step = 0.9, t_unit = 6.0, fix_us_per_tile = 0.13;  // 2 x 256 KiB at ~4 TB/s. (A sweep with LLMC_SYRK_S at
    // K = 4096, T = 262144 put S = 15 2 % ahead of the S = 9 this picks: within box-to-box noise, not adopted.)
    int smin = (int)(x_bytes / (1ll << 31)) + 1;
    if (smin > nk) smin = nk;
    if (smin < 1) smin = 1;
    int best = smin;
    double best_cost = 1e30;
    for (int S = smin; S <= 32 && S <= nk; ++S) {
        int64_t units = (int64_t)ntiles_real * S;
        double rounds = (double)ceil_div64(units, ncu);
        double cost = rounds * (((double)nk / S) * t_step + t_unit) + fix_us_per_tile * (double)units;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = S;
        }
    }
    return best;
}

}  // namespace llmc

using namespace llmc;

static int syrk_geometry(int64_t T, int64_t K, int64_t ldx, int* nb, int* ntp, int* S, int* nk) {
    *nb = (int)ceil_div64(K, TM);
    *ntp = tiles_padded(*nb);
    *nk = (int)ceil_div64(T, BK);
    int real = (*nb) * (*nb + 1) / 2;
    *S = choose_chunks(real, *nk, T * ldx * 2, 256);
    if (const char* e = getenv("LLMC_SYRK_S")) *S = atoi(e);   // diagnostic: force the token-chunk count
    if (*S > *nk) *S = *nk;
    if (*S < 1) *S = 1;
    return 0;
}

extern "C" size_t llmc_hessian_accum_ws_bytes(int64_t T, int64_t K, int64_t ldx) {
    if (T <= 0 || K <= 0 || ldx < K) return 0;
    int nb, ntp, S, nk;
    syrk_geometry(T, K, ldx, &nb, &ntp, &S, &nk);
    return (size_t)S * ntp * TILE_FLOATS * sizeof(float) + 256;
}

static int syrk_partials(const void* X, int dt, int64_t T, int64_t K, int64_t ldx, void* ws, hipStream_t st,
                         int* nb_o, int* ntp_o, int* S_o) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "hessian_accum: X must be f16 or bf16");
    LLMC_REQUIRE(X && ws && T > 0 && K > 0, "hessian_accum: null/empty argument");
    LLMC_REQUIRE(ldx >= K && ldx % 8 == 0 && ((uintptr_t)X & 15) == 0,
                 "hessian_accum: X rows must be 16-B aligned");
    LLMC_REQUIRE(K < (1 << 30), "hessian_accum: K too large");
    int nb, ntp, S, nk;
    syrk_geometry(T, K, ldx, &nb, &ntp, &S, &nk);
    SyrkArgs a;
    a.X = (const char*)X;
    a.T = T;
    a.ldx = ldx;
    a.K = (int)K;
    a.nb = nb;
    a.ntiles_p = ntp;
    a.S = S;
    a.nk = nk;
    a.part = (float*)ws;
    a.sync = (unsigned*)((char*)ws + (size_t)S * ntp * TILE_FLOATS * sizeof(float));
    a.kalign = SYRK_KALIGN;
    if (const char* e = getenv("LLMC_SYRK_KALIGN")) a.kalign = atoi(e) > 0 ? atoi(e) : SYRK_KALIGN;   // lab: 10 suits both rings
    if (getenv("LLMC_SYRK_NOSYNC")) a.sync = nullptr;
    if (a.sync) LLMC_HIP_CHECK(hipMemsetAsync(a.sync, 0, 4, st));
    // persistent: one workgroup per CU (a multiple of 8 keeps XCDs contiguous), minus the CUs the caller keeps free for
    // kernels of other streams (llmc_hip_set_cu_reserve): a k_syrk4 workgroup owns its CU, nothing co-resides with it
    int grid = (device_cu_count() - cu_reserve()) & ~7;
    if (grid < 8) grid = 8;
    // Kernel variant. Default k_syrk4 (one wave per SIMD, 128x128 wave tiles) with a 4-slot ring; LLMC_SYRK_V=5 its
    // 5-slot ring (all 160 KiB of LDS), =8 the 8-wave kernel, =2 its 4-stage ring, =88 its phase-split schedule.
    // All variants produce identical partial tiles for the same chunk alignment.
    int variant = SYRK_DEFAULT_VARIANT, abl = 0;
    if (const char* e = getenv("LLMC_SYRK_V")) variant = atoi(e);
    if (const char* e = getenv("LLMC_SYRK_ABL")) abl = atoi(e);   // lab only (wrong results by design)
    const bool bf = dt == LLMC_BF16;
    const void* fn = nullptr;
    int lds_bytes = SYRK_LDS, threads = SYRK_THREADS;
    if (variant == 2) {
        fn = bf ? (const void*)k_syrk2<LLMC_BF16> : (const void*)k_syrk2<LLMC_F16>;
        lds_bytes = SYRK2_LDS;
    } else if (variant == 88) {
        fn = bf ? (const void*)k_syrk<LLMC_BF16, true> : (const void*)k_syrk<LLMC_F16, true>;
    } else if (variant == 8) {
        fn = bf ? (const void*)k_syrk<LLMC_BF16, false> : (const void*)k_syrk<LLMC_F16, false>;
    } else if (variant == 5) {
        threads = S4_THREADS;
        lds_bytes = 5 * S4_STAGE;
        if (!bf) fn = (const void*)k_syrk4<LLMC_F16, 5, 0>;
        else if (abl == 4) fn = (const void*)k_syrk4<LLMC_BF16, 5, 4>;
        else if (abl == 1) fn = (const void*)k_syrk4<LLMC_BF16, 5, 1>;
        else fn = (const void*)k_syrk4<LLMC_BF16, 5, 0>;
    } else {
        threads = S4_THREADS;
        lds_bytes = 4 * S4_STAGE;
        if (!bf) fn = (const void*)k_syrk4<LLMC_F16, 4, 0>;
        else if (abl == 1) fn = (const void*)k_syrk4<LLMC_BF16, 4, 1>;
        else if (abl == 2) fn = (const void*)k_syrk4<LLMC_BF16, 4, 2>;
        else if (abl == 3) fn = (const void*)k_syrk4<LLMC_BF16, 4, 3>;
        else if (abl == 4) fn = (const void*)k_syrk4<LLMC_BF16, 4, 4>;
        else if (abl == 8) fn = (const void*)k_syrk4<LLMC_BF16, 4, 8>;
        else if (abl == 19) fn = (const void*)k_syrk4<LLMC_BF16, 4, 19>;
        else if (abl == 18) fn = (const void*)k_syrk4<LLMC_BF16, 4, 18>;
        else fn = (const void*)k_syrk4<LLMC_BF16, 4, 0>;
    }
    LLMC_REQUIRE(a.kalign % 2 == 0 || variant == 5, "hessian_accum: chunk alignment must be even for 4-slot rings");
    LLMC_REQUIRE(variant != 5 || a.kalign % 5 == 0, "hessian_accum: the 5-slot ring needs LLMC_SYRK_KALIGN % 5 == 0");
    int rc = ensure_dynamic_lds(fn, lds_bytes);
    if (rc) return rc;
    void* kargs[] = {(void*)&a};
    LLMC_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(threads), kargs, (size_t)lds_bytes, st));
    LLMC_LAUNCH_CHECK();
    *nb_o = nb;
    *ntp_o = ntp;
    *S_o = S;
    return LLMC_OK;
}

extern "C" int llmc_hessian_accum_partials(const void* X, int dt, int64_t T, int64_t K, int64_t ldx, void* ws,
                                           llmc_stream_t stream) {
    int nb, ntp, S;
    return syrk_partials(X, dt, T, K, ldx, ws, (hipStream_t)stream, &nb, &ntp, &S);
}

extern "C" int llmc_hessian_accum_reduce(float* H, int64_t T, int64_t K, int64_t ldx, double n_before,
                                         double n_after, const void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(H && ws && T > 0 && K > 0 && n_after > 0, "hessian_accum_reduce: bad argument");
    int nb, ntp, S, nk;
    syrk_geometry(T, K, ldx, &nb, &ntp, &S, &nk);
    float alpha = (float)(n_before / n_after);
    float beta = (float)(2.0 / n_after);
    hipLaunchKernelGGL(k_syrk_fixup, dim3(TILE_FLOATS / 4 / 256, ntp), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, H, (int)K, nb, ntp, S, alpha, beta);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_hessian_accum(float* H, const void* X, int dt, int64_t T, int64_t K, int64_t ldx,
                                  double n_before, double n_after, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(H != nullptr, "hessian_accum: null H");
    int rc = llmc_hessian_accum_partials(X, dt, T, K, ldx, ws, stream);
    if (rc) return rc;
    return llmc_hessian_accum_reduce(H, T, K, ldx, n_before, n_after, ws, stream);
}
