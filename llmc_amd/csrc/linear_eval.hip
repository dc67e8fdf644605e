// linear_eval.hip — K9: the fake-quant W4A16 matmul of AWQ's scale search (awq.py:110-145, 229-236):
//   Y = X [N, K] . Wq [R, K]^T   16-bit operands, fp32 accumulation on v_mfma_f32_32x32x16, Y rounded to the
//   model dtype exactly once (what F.linear returns), then either stored (get_original_out) or compared
//   with the stored original output: loss = sum((Y0 - Y)^2), the difference formed in the model dtype like
//   the reference's `(org_out - out).float().pow(2)`.
// Both operands are contiguous along the contraction axis ("NT"): tiles are staged row-major by LDS-DMA and
// read with ds_read_b128; the 16-B chunk index is XOR-ed with ((row >> 1) & 7) on the DMA source address
// and on the read, which makes every ds_read_b128 lane group hit 16 distinct 16-B bank slots.
// Two kernels: k_linear_eval (row-major operands: tile 256 x 256, K-step 64, 8 waves as 2 x 4; what FakeQuantLinear.forward
// and any K % 128 != 0 shape run on) and k_linear_eval4 (k-tiled operands, one wave per SIMD: the AWQ grid, and K3's deep
// products through gemm6 at the end of the file). Persistent grids; each XCD works on one block of tiles per round
// (lin_tile_order), so the workgroups that share an L2 share few operand panels at the same K position.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "mfma_common.h"
#include "sgemm.h"

namespace llmc {

static constexpr int LT = 256;
static constexpr int LBK = 64;
static constexpr int LPANEL = LT * LBK * 2;   // 32 KiB
static constexpr int LSTAGE = 2 * LPANEL;
static constexpr int LLDS = 2 * LSTAGE;       // 128 KiB
static constexpr int LTHREADS = 512;

struct LinArgs {
    const char* X;    // [N, K]
    const char* W;    // [R, K]
    int64_t N, K, R;
    int mode;         // 0 store Y, 1 loss vs Y0
    char* Y;          // [N, R] dt (mode 0)
    const char* Y0;   // [N, R] dt (mode 1); optional bias [R] dt (mode 0)
    float* part;      // [ntiles] partial loss sums (mode 1)
    int ntm, ntn;
    // tile order: each XCD (blockIdx & 7) works on one sbm x sbn block of tiles per round (sbm * sbn = gridDim / 8), so
    // the CUs that share an L2 share sbm + sbn operand panels instead of ~gridDim/8 + 1; blocks are numbered tn-fastest
    int sbm, sbn, nsn, nrounds;
    int yblk;         // k_linear_eval4: Y (mode 0) / Y0 (mode 1) is tile-blocked in fragment order (LLMC_LINEAR_YBLOCKED)
    int y0_lds;       // k_linear_eval4, mode 1: Y0 tiles are staged through LDS (R % 8 == 0, Y0 16-B aligned, < 4 GiB)
    // mode 3 (gemm6, K3's large products): C fp32 [N, R] window with row stride ldc = csign * (X W^T); krange limits a
    // tile's k loop: 1 = op(A) upper triangular (k >= first row of the tile), 2 = op(B) upper triangular (k < last column of
    // the tile + 1), in units of kunit k per source k (the stacked planes: 6)
    float* C;
    int64_t ldc;
    float csign;
    int krange, kunit;
    unsigned* queue;  // gemm6: tile counter (zeroed before the launch), or null = the static XCD block order
    // split-K (k_linear_eval4, mode 3 into `C`): ntm = ntm_real * ksplit tile rows; tile row tm' is k-slice tm' / ntm_real of
    // tile row tm' % ntm_real and writes its fp32 partial into C + slice * N * ldc (summed by k_splitk_reduce). ksplit <= 1: off
    int ksplit, ntm_real;
};

// (round, block) -> tile; false = padding of a ragged block
__device__ __forceinline__ bool lin_tile(const LinArgs& a, int round, int& tm, int& tn) {
    const int slot = blockIdx.x >> 3;
    const int si = slot / a.sbn, sj = slot - si * a.sbn;
    const int sid = round * 8 + (blockIdx.x & 7);
    const int sm = sid / a.nsn, sn = sid - sm * a.nsn;
    tm = sm * a.sbm + si;
    tn = sn * a.sbn + sj;
    return tm < a.ntm && tn < a.ntn;
}

static void lin_tile_order(LinArgs& a, int grid) {
    const int spx = grid / 8;
    int best = 1;
    int64_t best_cost = -1;
    for (int sbn = 1; sbn <= 8; sbn *= 2) {
        if (spx % sbn) continue;
        const int sbm = spx / sbn;
        const int64_t padded = ceil_div64(a.ntm, sbm) * sbm * ceil_div64(a.ntn, sbn) * sbn;
        const int64_t cost = padded * 64 + (sbm + sbn);   // fewest padded tiles, then fewest panels per XCD
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sbn; }
    }
    a.sbn = best;
    a.sbm = spx / best;
    a.nsn = (int)ceil_div64(a.ntn, a.sbn);
    a.nrounds = (int)ceil_div64(ceil_div64(a.ntm, a.sbm) * a.nsn, 8);
}

template <int DT>
__global__ __launch_bounds__(LTHREADS) void k_linear_eval(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)(smem + LLDS);  // 8 floats behind the stages (one LDS object: G17)
    LDS_AS char* lds = (LDS_AS char*)smem;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    const int64_t row_bytes = a.K * 2;

    // DMA: instruction q of wave wv fills KiB-block (q*8 + wv) of a panel = rows 8*blk .. 8*blk+7
    const int drow = lane >> 3;                        // row inside the 8-row piece
    const int dsw = (4 * (wv & 1) + (lane >> 4)) & 7;  // ((row >> 1) & 7) for this lane's row
    const int dlc = (lane & 7) ^ dsw;                  // logical 16-B chunk stored in this physical slot
    const uint32_t voff0 = (uint32_t)((int64_t)(8 * wv + drow) * row_bytes + dlc * 16);
    const uint32_t slab = (uint32_t)(64 * row_bytes);  // 64 rows per instruction index q

    // fragment reads: row = base + (lane & 31), logical chunk = 2*kk + (lane >> 5)
    const int rsw = (lane >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (((2 * kk + (lane >> 5)) ^ rsw) << 4);
    const int rowA = (wm * 128 + (lane & 31)) * (LBK * 2);
    const int rowB = (wn * 64 + (lane & 31)) * (LBK * 2);

    i32x4 rx, rw;
    {
        const int64_t xb = a.N * row_bytes, wb = a.R * row_bytes;
        rx[0] = (int)(uint32_t)(uintptr_t)a.X;
        rx[1] = (int)((uint32_t)((uintptr_t)a.X >> 32) & 0xffffu);
        rx[2] = (int)(uint32_t)(xb > 0xffffffffll ? 0xffffffffll : xb);
        rx[3] = 0x00020000;
        rw[0] = (int)(uint32_t)(uintptr_t)a.W;
        rw[1] = (int)((uint32_t)((uintptr_t)a.W >> 32) & 0xffffu);
        rw[2] = (int)(uint32_t)(wb > 0xffffffffll ? 0xffffffffll : wb);
        rw[3] = 0x00020000;
    }

    const int nk = (int)(a.K / LBK);

    for (int round = 0; round < a.nrounds; ++round) {
        int tm, tn;
        if (!lin_tile(a, round, tm, tn)) continue;
        const int t = tm * a.ntn + tn;
        const uint32_t baseX = (uint32_t)((int64_t)tm * LT * row_bytes) + voff0;
        const uint32_t baseW = (uint32_t)((int64_t)tn * LT * row_bytes) + voff0;

        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        auto stage = [&](int buf, int ks) {
            const uint32_t koffb = (uint32_t)(ks * LBK * 2);
            const uint32_t dst = lds_base + buf * LSTAGE + wv * 1024;
#pragma unroll
            for (int q = 0; q < 4; ++q) dma16(rx, baseX + koffb + q * slab, dst + q * 8192);
#pragma unroll
            for (int q = 0; q < 4; ++q) dma16(rw, baseW + koffb + q * slab, dst + LPANEL + q * 8192);
        };

        int cur = 0;
        stage(0, 0);
        for (int ks = 0; ks < nk; ++ks) {
            dma_wait_all();
            __syncthreads();
            if (ks + 1 < nk) stage(cur ^ 1, ks + 1);
            LDS_AS char* pa = lds + cur * LSTAGE + rowA;
            LDS_AS char* pb = lds + cur * LSTAGE + LPANEL + rowB;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s16x8 fa[4], fb[2];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    fa[m] = *(LDS_AS s16x8*)(pa + m * 32 * (LBK * 2) + koff[kk]);
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    fb[n] = *(LDS_AS s16x8*)(pb + n * 32 * (LBK * 2) + koff[kk]);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = Mfma<DT>::run(fa[m], fb[n], acc[m][n]);
            }
            cur ^= 1;
        }
        __syncthreads();

        // epilogue: acc[m][n][r] -> token = tm*256 + wm*128 + m*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
        //                           out col = tn*256 + wn*64 + n*32 + (lane & 31)
        float lsum = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int64_t col = (int64_t)tn * LT + wn * 64 + n * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t tok = (int64_t)tm * LT + wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (tok < a.N && col < a.R) {
                        if (a.mode == 0) {
                            // optional bias [R] rides in Y0: added to the fp32 sum before the single rounding (addmm)
                            const float b = a.Y0 ? load_as_f32(a.Y0, col, DT) : 0.0f;
                            store_from_f32(a.Y, tok * a.R + col, DT, rndc<DT>(acc[m][n][r] + b));
                        } else {
                            const float y = rndc<DT>(acc[m][n][r]);
                            const float y0 = load_as_f32(a.Y0, tok * a.R + col, DT);
                            const float d = rndc<DT>(y0 - y);
                            lsum += d * d;
                        }
                    }
                }
            }
        if (a.mode == 1) {
            lsum = wave_sum(lsum, 64);
            if (lane == 0) red[wv] = lsum;
            __syncthreads();
            if (tid == 0) {
                float s = 0.0f;
                for (int i = 0; i < LTHREADS / 64; ++i) s += red[i];
                a.part[t] = s;
            }
            __syncthreads();
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// k_linear_eval4 — the same product from K-TILED operands, with hessian_syrk.hip's one-wave-per-SIMD structure (k_syrk4).
// Operand layout ("kt"): T[K/32][rows][32] — the 32-k slice of every row contiguous, so the 256 x 32 panel of one stage
// is ONE contiguous 16-KiB run and every LDS-DMA piece reads eight whole 128-B lines. (Read row-major, a 32-k stage
// touches half a line per row and the other half one stage = 32 KiB of traffic later, after the 32-KiB vector L1 has
// turned over: twice the L2->L1 traffic, measured 0.34 of peak against the 8-wave kernel's 0.38.) llmc_ktile_pack
// (below) makes the layout; inside the AWQ search the producers write it directly.
// 4 waves as 2 x 2, each 128 x 128 = 4 x 4 accumulators (256 registers in the AGPR half), 8 fragment reads per 16
// MFMAs, LDS ring of 4 stages of 32 k, three stages of LDS-DMA in flight with a counted vmcnt, fragments
// double-buffered in registers, one s_barrier per stage between its two MFMA bursts, ring slots compile-time (stage
// loop unrolled by 4, so K % 128 == 0), behind every MFMA at most one fragment read (ONE ds_read_b128: both operands
// are k-contiguous) or one LDS-DMA piece. LDS image of a panel: [256 rows][64 B]; the 16-B chunk index is XOR-ed with
// ((row >> 2) & 3) on the DMA source address and on the read: the four rows of a ds_read_b128 lane group that share a
// bank column sit in four chunks. The k order of the sum is k_linear_eval's (ascending, 16 per MFMA): same bits.
// -----------------------------------------------------------------------------------------------------------------
static constexpr int L4_THREADS = 256;
static constexpr int L4_KS = 32;                       // k per stage
static constexpr int L4_PANEL = LT * L4_KS * 2;        // 16 KiB
static constexpr int L4_STAGE = 2 * L4_PANEL;
static constexpr int L4_RING = 4;
static constexpr int L4_LDS = L4_RING * L4_STAGE;      // 128 KiB

template <int I, int N, typename F> __device__ __forceinline__ void l4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        l4_static_for<I + 1, N>(f);
    }
}

template <int DST>
__device__ __forceinline__ void l4_dma(i32x4 rsrc, uint32_t voff, uint32_t& soff, uint32_t wvoff, uint32_t adv) {
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\ts_add_u32 %0, %0, %5"
                 : "+s"(soff) : "v"(voff), "s"(rsrc), "s"(wvoff), "n"(DST), "s"(adv) : "memory", "scc");
}
template <int DST>
__device__ __forceinline__ void l4_dma_at(i32x4 rsrc, uint32_t voff, uint32_t soff, uint32_t wvoff) {
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
                 :: "s"(soff), "v"(voff), "s"(rsrc), "s"(wvoff), "n"(DST) : "memory", "scc");
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// (a, b) rounded to bf16 (RNE, one v_cvt_pk_bf16_f32) and widened back
__device__ __forceinline__ f32x2 round_pair_bf16(f32x2 v) {
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
template <int N> __device__ __forceinline__ void l4_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// STAGE_Y: loss mode with the Y0 tile staged through LDS: 1 row-major Y0 (a.y0_lds), 2 tile-blocked Y0 (a.yblk)
template <int DT, int STAGE_Y>
__global__ __launch_bounds__(L4_THREADS) void k_linear_eval4(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)(smem + L4_LDS);
    LDS_AS char* lds = (LDS_AS char*)smem;
    if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // DMA destinations are immediates
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int64_t row_bytes = a.K * 2;
    constexpr int PER = 8;

    // DMA piece q (0..3) of wave wv = KiB-block (q*4 + wv) of a panel = rows 16*blk .. 16*blk + 15, 4 lanes per row
    const int drow = lane >> 2;
    const int dlog = (lane & 3) ^ ((lane >> 4) & 3);                    // logical chunk held by this physical slot
    const uint32_t vlane = (uint32_t)((wv * 16 + drow) * 64 + dlog * 16);
    const uint32_t slab = 64u * 64u;                                    // 64 rows per piece index q
    const uint32_t advA = (uint32_t)(a.N * 64), advB = (uint32_t)(a.R * 64);   // one 32-k slice of every row
    const uint32_t wvoff = (uint32_t)wv * 1024u;

    // fragment addresses: slot j at j * 32 KiB; one base per pair of slots (+ immediate 0 / 32 KiB)
    int offA[2][2][4], offB[2][2][4];                                  // [slot pair][kk][frag]
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ra = wm * 128 + m * 32 + (lane & 31), rb = wn * 128 + m * 32 + (lane & 31);
                const int cl = 2 * kk + (lane >> 5);
                offA[pr][kk][m] = pr * 2 * L4_STAGE + ra * 64 + ((cl ^ ((ra >> 2) & 3)) << 4);
                offB[pr][kk][m] = pr * 2 * L4_STAGE + L4_PANEL + rb * 64 + ((cl ^ ((rb >> 2) & 3)) << 4);
            }

    i32x4 rx, rw;
    {
        const int64_t xb = a.N * row_bytes, wb = a.R * row_bytes;
        rx[0] = (int)(uint32_t)(uintptr_t)a.X;
        rx[1] = (int)((uint32_t)((uintptr_t)a.X >> 32) & 0xffffu);
        rx[2] = (int)(uint32_t)(xb > 0xffffffffll ? 0xffffffffll : xb);
        rx[3] = 0x00020000;
        rw[0] = (int)(uint32_t)(uintptr_t)a.W;
        rw[1] = (int)((uint32_t)((uintptr_t)a.W >> 32) & 0xffffu);
        rw[2] = (int)(uint32_t)(wb > 0xffffffffll ? 0xffffffffll : wb);
        rw[3] = 0x00020000;
    }
    i32x4 ry = rx;
    if (a.mode == 1) {
        const int64_t yb = a.yblk ? (int64_t)a.ntm * a.ntn * 131072 : a.N * a.R * 2;
        ry[0] = (int)(uint32_t)(uintptr_t)a.Y0;
        ry[1] = (int)((uint32_t)((uintptr_t)a.Y0 >> 32) & 0xffffu);
        ry[2] = (int)(uint32_t)(yb > 0xffffffffll ? 0xffffffffll : yb);
    }
    for (int round = 0; round < a.nrounds; ++round) {
        int tm, tn;
        if (a.queue) {
            // gemm6: tiles differ in depth (triangular operand), so workgroups pull them from a counter, deepest first
            // (a_upper: small tm first; b_upper: large tn first); the 256 tiles in flight still share few operand panels
            __syncthreads();
            if (tid == 0) *(int*)red = (int)atomicAdd(a.queue, 1u);
            __syncthreads();
            const int q = __builtin_amdgcn_readfirstlane(*(volatile int*)red);
            if (q >= a.ntm * a.ntn) break;
            if (a.krange == 2) { tn = a.ntn - 1 - q / a.ntm; tm = q - (q / a.ntm) * a.ntm; }
            else { tm = q / a.ntn; tn = q - tm * a.ntn; }
            round = -1;     // the loop ends by the break above
        } else if (!lin_tile(a, round, tm, tn)) continue;
        int ks = 0;                               // split-K: which k-slice of the tile this unit is
        if (a.ksplit > 1) {
            ks = __builtin_amdgcn_readfirstlane(tm / a.ntm_real);    // (the division goes through the VALU; the value is uniform)
            tm -= ks * a.ntm_real;
        }
        const int t = tm * a.ntn + tn;
        // triangular operands (gemm6): stages [st0, st1) of 32 k; both ends are multiples of a ring turn
        int st0 = 0, st1 = (int)(a.K / L4_KS);
        if (a.krange == 1) st0 = tm * LT * a.kunit / L4_KS;
        if (a.krange == 2) st1 = min(st1, (tn + 1) * LT * a.kunit / L4_KS);
        if (a.ksplit > 1) {                       // equal slices of whole ring turns, the last one takes the remainder
            const int per = ((st1 / a.ksplit) >> 2) << 2;
            st0 = ks * per;
            if (ks + 1 < a.ksplit) st1 = st0 + per;
        }
        const int ngroups = (st1 - st0) / L4_RING;
        if (ngroups <= 0 && a.mode != 3) continue;
        const uint32_t vA = (uint32_t)(tm * LT * 64) + vlane;
        const uint32_t vB = (uint32_t)(tn * LT * 64) + vlane;
        f32x16 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
        uint32_t sA[4], sB[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sA[q] = (uint32_t)q * slab + (uint32_t)st0 * advA;
            sB[q] = (uint32_t)q * slab + (uint32_t)st0 * advB;
        }
        auto piece = [&](auto slc, auto dc) {
            constexpr int SL = decltype(slc)::value;
            constexpr int d = decltype(dc)::value;
            constexpr int DSTB = SL * L4_STAGE + (d >> 2) * L4_PANEL + (d & 3) * 4 * 1024;
            if constexpr (d < 4) l4_dma<DSTB>(rx, vA, sA[d & 3], wvoff, advA);
            else l4_dma<DSTB>(rw, vB, sB[d & 3], wvoff, advB);
        };
        const uint32_t vy = (uint32_t)((((int64_t)tm * LT + 2 * wv + (lane >> 5)) * a.R + (int64_t)tn * LT + (lane & 31) * 8) * 2);
        const uint32_t advY = (uint32_t)(16 * a.R);        // 8 rows down per piece index
        const uint32_t vyb = (uint32_t)(((int64_t)t * 4 + wv) * 32768 + lane * 16);
        s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
        auto frag = [&](auto slc, auto kkc, auto fc, s16x8 (&fa)[4], s16x8 (&fb)[4]) {
            constexpr int SL = decltype(slc)::value;
            constexpr int KK = decltype(kkc)::value;
            constexpr int f = decltype(fc)::value;
            constexpr int IMM = (SL & 1) * L4_STAGE;
            if constexpr (f == 0) fa[0] = *(LDS_AS s16x8*)(lds + offA[SL >> 1][KK][0] + IMM);
            else if constexpr (f <= 4) fb[f - 1] = *(LDS_AS s16x8*)(lds + offB[SL >> 1][KK][f - 1] + IMM);
            else fa[f - 4] = *(LDS_AS s16x8*)(lds + offA[SL >> 1][KK][f - 4] + IMM);
        };
        // Y0 tile piece I (0..31) of this wave. Row-major Y0: rows 2 * (wv + 4 I), + 1 (512 B each) -> LDS
        // [I * 4 KiB + wv KiB). Tile-blocked Y0 (fragment order, LLMC_LINEAR_YBLOCKED): the wave's own KiB I of the
        // tile's contiguous 128 KiB -> LDS [I/8 * 32 KiB + wv * 8 KiB + I%8 KiB): 8 pieces of every wave per ring slot
        auto ypiece = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (STAGE_Y == 2) l4_dma_at<(I >> 3) * 32768 + (I & 7) * 1024>(ry, vyb, (uint32_t)I * 1024u, wvoff * 8);
            else l4_dma_at<I * 4096>(ry, vy, (uint32_t)I * advY, wvoff);
        };
        // REQ: what rides behind the MFMAs. 0 nothing, 1 operand pieces D0.. of slot dslc (the steady state),
        // 2 Y0 pieces D0..D0+3, 3 Y0 pieces D0..D0+7 and no fragment reads (the tile's very last burst)
        auto burst = [&](const s16x8 (&fa)[4], const s16x8 (&fb)[4], s16x8 (&na)[4], s16x8 (&nb)[4], auto rslc, auto kkc,
                         auto dslc, auto d0c, auto reqc) {
            constexpr int D0 = decltype(d0c)::value;
            constexpr int REQ = decltype(reqc)::value;
            l4_static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int mi = i >> 2, ni = (mi & 1) ? 3 - (i & 3) : (i & 3);
                acc[mi][ni] = Mfma<DT>::run(fa[mi], fb[ni], acc[mi][ni]);
                if constexpr ((i & 3) == 3) {
                    if constexpr (REQ == 1) piece(dslc, std::integral_constant<int, D0 + (i >> 2)>{});
                    else if constexpr (REQ >= 2) ypiece(std::integral_constant<int, D0 + (i >> 2)>{});
                } else if constexpr (REQ == 3) {
                    if constexpr (i - (i >> 2) < 4) ypiece(std::integral_constant<int, D0 + 4 + i - (i >> 2)>{});
                } else if constexpr (i < 10) frag(rslc, kkc, std::integral_constant<int, i - (i >> 2)>{}, na, nb);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        constexpr std::integral_constant<int, 0> K0{};
        constexpr std::integral_constant<int, 1> K1{};
        if (ngroups > 0) {
        l4_static_for<0, L4_RING - 1>([&](auto slc) { l4_static_for<0, 8>([&](auto dc) { piece(slc, dc); }); });
        l4_static_for<0, 4>([&](auto dc) { piece(std::integral_constant<int, L4_RING - 1>{}, dc); });
        l4_vmwait<(L4_RING - 2) * PER + PER / 2>();
        __builtin_amdgcn_s_barrier();
        l4_static_for<0, 8>([&](auto fc) { frag(std::integral_constant<int, 0>{}, K0, fc, fa0, fb0); });
        }
        // one ring turn = 4 stages. The LAST turn of a tile (KIND 1) requests only what the tile still needs (the B half
        // of its final stage) and counts the VM counter down to 0, so the ring is quiet and free when the turn ends.
        // KIND 2 (loss mode): the ring slots the last turn leaves behind take the Y0 tile, a quarter (64 rows = 32 KiB =
        // one slot = 8 pieces per wave) per stage, in the request slots the operands no longer use: the loss epilogue
        // finds its reference tile in LDS.
        auto turn = [&](auto kindc) {
            constexpr int KIND = decltype(kindc)::value;
            l4_static_for<0, L4_RING>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                constexpr int JN = (J + 1) % L4_RING, JP = (J + L4_RING - 1) % L4_RING;
                // requests that may still be in flight when stage J + 1 must have landed
                constexpr int AHEAD = KIND == 0 ? 2 * PER : KIND == 1 ? (J < 2 ? 2 - J : 0) * PER : 2 * PER;
                constexpr int R1 = KIND == 0 ? 1 : J == 0 ? 1 : KIND == 1 ? 0 : 2;
                constexpr int R2 = KIND == 0 ? 1 : KIND == 1 ? 0 : J == 3 ? 3 : 2;
                burst(fa0, fb0, fa1, fb1, jc, K1, std::integral_constant<int, JP>{},
                      std::integral_constant<int, R1 == 1 ? 4 : 8 * J - 4>{}, std::integral_constant<int, R1>{});
                if constexpr (!(KIND == 2 && J == 3)) l4_vmwait<AHEAD>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                burst(fa1, fb1, fa0, fb0, std::integral_constant<int, JN>{}, K0, jc,
                      std::integral_constant<int, R2 == 1 ? 0 : 8 * J>{}, std::integral_constant<int, R2>{});
            });
        };
        for (int g = 0; g < ngroups - 1; ++g) turn(std::integral_constant<int, 0>{});
        if (ngroups > 0) turn(std::integral_constant<int, STAGE_Y != 0 ? 2 : 1>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef LLMC_LAB
        if (a.mode == 2) {   // lab (LLMC_LIN_ABL=1): the main loop alone, nothing stored
            __builtin_amdgcn_s_barrier();
            continue;
        }
#endif

        if constexpr (STAGE_Y != 0) {
            // loss epilogue: the Y0 tile (256 x 512 B) arrived in the ring during the last turn; every lane picks its 256
            // values with immediate-offset ds_read_u16. (Read straight from global, each of the 256 two-byte loads sat
            // behind its own s_waitcnt: ~40 % of the tile time.)
            l4_vmwait<0>();
            if constexpr (STAGE_Y == 1) __builtin_amdgcn_s_barrier();   // blocked: a wave reads only what it requested itself
#ifdef LLMC_LAB
            if (a.y0_lds == 2) { __syncthreads(); continue; }   // lab (LLMC_LIN_ABL=3): Y0 staged and waited for, not folded
#endif
            const int ybase = (wm * 128 + 4 * (lane >> 5)) * 512 + (wn * 128 + (lane & 31)) * 2;
            const int yown = wv * 8192 + lane * 16;
            const int64_t col0 = (int64_t)tn * LT + wn * 128 + (lane & 31);
            const int64_t tokb = (int64_t)tm * LT + wm * 128 + 4 * (lane >> 5);
            // rows / columns of this lane's 128 x 128 block that exist (edge tiles); interior tiles take the unmasked copy
            const int nrow = (int)(a.N - tokb < 128 ? a.N - tokb : 128);      // valid: trow < nrow (may be <= 0)
            const int ncol = (int)(a.R - col0 <= 0 ? 0 : (a.R - col0 + 31) / 32);   // valid: n < ncol
            const bool full = (int64_t)(tm + 1) * LT <= a.N && (int64_t)(tn + 1) * LT <= a.R;   // block-uniform
            f32x2 lsum2 = {0.0f, 0.0f};      // even / odd accumulator registers, fused multiply-add in fp32
            auto fold = [&](auto maskedc) {
                constexpr bool MASKED = decltype(maskedc)::value;
                l4_static_for<0, 16>([&](auto mnc) {
                    constexpr int m = decltype(mnc)::value >> 2, n = decltype(mnc)::value & 3;
                    uint16_t yb[16];
                    if constexpr (STAGE_Y == 2) {
                        constexpr int I = (m * 4 + n) * 2;
                        const s16x8 v0 = *(LDS_AS const s16x8*)(lds + yown + (I >> 3) * 32768 + (I & 7) * 1024);
                        const s16x8 v1 = *(LDS_AS const s16x8*)(lds + yown + (I >> 3) * 32768 + ((I & 7) + 1) * 1024);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            yb[j] = (uint16_t)v0[j];
                            yb[8 + j] = (uint16_t)v1[j];
                        }
                    } else {
                        l4_static_for<0, 16>([&](auto rc) {
                            constexpr int r = decltype(rc)::value;
                            constexpr int trow = m * 32 + (r & 3) + 8 * (r >> 2);
                            yb[r] = *(LDS_AS const uint16_t*)(lds + ybase + trow * 512 + n * 64);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);   // 16 reads in flight, one wait (the scheduler pairs each read with its use)
                    // two values per step: one packed conversion per rounding, packed fp32 subtract / multiply-add
                    l4_static_for<0, 8>([&](auto pc) {
                        constexpr int r = 2 * decltype(pc)::value;
                        constexpr int trow = m * 32 + (r & 3) + 8 * (r >> 2);      // r + 1 is the next row
                        f32x2 y0, y, d;
                        if constexpr (DT == LLMC_BF16) {
                            y0 = f32x2{bf16_bits_to_f32(yb[r]), bf16_bits_to_f32(yb[r + 1])};
                            y = round_pair_bf16(f32x2{acc[m][n][r], acc[m][n][r + 1]});
                            d = round_pair_bf16(y0 - y);
                        } else {
                            y0 = f32x2{f16_bits_to_f32(yb[r]), f16_bits_to_f32(yb[r + 1])};
                            y = f32x2{rndc<DT>(acc[m][n][r]), rndc<DT>(acc[m][n][r + 1])};
                            const f32x2 t = y0 - y;
                            d = f32x2{rndc<DT>(t[0]), rndc<DT>(t[1])};
                        }
                        if constexpr (MASKED) {
                            // bit mask, not a select: the compiler turns `cond ? d2 : 0` into exec-masked branches
                            const uint32_t k0 = (uint32_t)-(int)((trow < nrow) & (n < ncol));
                            const uint32_t k1 = (uint32_t)-(int)((trow + 1 < nrow) & (n < ncol));
                            d = f32x2{__uint_as_float(__float_as_uint(d[0]) & k0), __uint_as_float(__float_as_uint(d[1]) & k1)};
                        }
                        lsum2 = __builtin_elementwise_fma(d, d, lsum2);
                    });
                });
            };
            if (full) fold(std::false_type{});
            else fold(std::true_type{});
            float lsum = lsum2[0] + lsum2[1];
            lsum = wave_sum(lsum, 64);
            if (lane == 0) red[wv] = lsum;
            __syncthreads();           // also: every wave is done with the staged tile before the next prologue
            if (tid == 0) a.part[t] = (red[0] + red[1]) + (red[2] + red[3]);
            __syncthreads();
            continue;
        }

        __builtin_amdgcn_s_barrier();    // every wave has read its last fragments before the next prologue refills the ring
        if (a.mode == 3) {
            // gemm6 / split-K: C = csign * acc in fp32; a lane's 32 consecutive columns per row segment are one 128-B line
            float* const Cs = a.C + (int64_t)ks * a.N * a.ldc;
            l4_static_for<0, 16>([&](auto mnc) {
                constexpr int m = decltype(mnc)::value >> 2, n = decltype(mnc)::value & 3;
                const int64_t col = (int64_t)tn * LT + wn * 128 + n * 32 + (lane & 31);
                const int64_t tok0 = (int64_t)tm * LT + wm * 128 + m * 32 + 4 * (lane >> 5);
                if (col < a.R) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t tok = tok0 + (r & 3) + 8 * (r >> 2);
                        if (tok < a.N) Cs[tok * a.ldc + col] = a.csign * acc[m][n][r];
                    }
                }
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        if (a.mode == 0 && a.yblk) {
            // tile-blocked output in fragment order: tile t = 128 KiB, wave wv = 32 KiB, piece I = (m*4 + n)*2 + v = 1 KiB,
            // lane = 16 B = acc[m][n][8v .. 8v+7] rounded (+ bias): one coalesced 16-B store per piece
            char* yt = a.Y + ((int64_t)t * 4 + wv) * 32768 + lane * 16;
            l4_static_for<0, 16>([&](auto mnc) {
                constexpr int m = decltype(mnc)::value >> 2, n = decltype(mnc)::value & 3;
                const int64_t col = (int64_t)tn * LT + wn * 128 + n * 32 + (lane & 31);
                const float b = (a.Y0 && col < a.R) ? load_as_f32(a.Y0, col, DT) : 0.0f;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    uint16_t o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = acc[m][n][8 * v + j] + b;
                        o[j] = DT == LLMC_BF16 ? f32_to_bf16_bits(y) : f32_to_f16_bits(y);
                    }
                    uint4 ov;
                    __builtin_memcpy(&ov, o, 16);
                    *reinterpret_cast<uint4*>(yt + ((m * 4 + n) * 2 + v) * 1024) = ov;
                }
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        if (a.mode == 0 && (a.R & 7) == 0 && (((uintptr_t)a.Y) & 15) == 0) {
            // row-major output through the (now idle) operand ring: a wave lays its 128 x 128 results out row by row in its own
            // 32 KiB (2-byte LDS writes, the accumulator layout has a column per lane), then stores 16 bytes per lane = four whole
            // 256-byte row segments per instruction. Two-byte global stores straight from the accumulators cost 0.65 ms on a
            // 65536 x 4096 output (0.90 against 1.22 PFLOP/s for the tile-blocked form, profiles/r04_linear_paths.txt).
            LDS_AS char* my = lds + wv * 32768;
            l4_static_for<0, 16>([&](auto mnc) {
                constexpr int m = decltype(mnc)::value >> 2, n = decltype(mnc)::value & 3;
                const int64_t col = (int64_t)tn * LT + wn * 128 + n * 32 + (lane & 31);
                const float b = (a.Y0 && col < a.R) ? load_as_f32(a.Y0, col, DT) : 0.0f;
                l4_static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float y = acc[m][n][r] + b;
                    *(LDS_AS uint16_t*)(my + row * 256 + (n * 32 + (lane & 31)) * 2) =
                        DT == LLMC_BF16 ? f32_to_bf16_bits(y) : f32_to_f16_bits(y);
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own writes, before its own reads
            const int64_t tok0 = (int64_t)tm * LT + wm * 128, col0 = (int64_t)tn * LT + wn * 128;
#pragma unroll 2
            for (int i = 0; i < 32; ++i) {
                const int id = i * 64 + lane, row = id >> 4, c16 = id & 15;
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = *(LDS_AS u32x4_t*)(my + row * 256 + c16 * 16);
                const int64_t tok = tok0 + row, col = col0 + c16 * 8;
                if (tok < a.N && col < a.R) *reinterpret_cast<u32x4_t*>(a.Y + (tok * a.R + col) * 2) = v;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // every wave is done with the ring before the next prologue refills it
            continue;
        }
        float lsum = 0.0f;
        l4_static_for<0, 16>([&](auto mnc) {
            constexpr int m = decltype(mnc)::value >> 2, n = decltype(mnc)::value & 3;
            const int64_t col = (int64_t)tn * LT + wn * 128 + n * 32 + (lane & 31);
            const int64_t tok0 = (int64_t)tm * LT + wm * 128 + m * 32 + 4 * (lane >> 5);
            l4_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int64_t tok = tok0 + (r & 3) + 8 * (r >> 2);
                if (tok < a.N && col < a.R) {
                    if (a.mode == 0) {
                        const float b = a.Y0 ? load_as_f32(a.Y0, col, DT) : 0.0f;
                        store_from_f32(a.Y, tok * a.R + col, DT, rndc<DT>(acc[m][n][r] + b));
                    } else {
                        const float y = rndc<DT>(acc[m][n][r]);
                        const float y0 = load_as_f32(a.Y0, tok * a.R + col, DT);
                        const float d = rndc<DT>(y0 - y);
                        lsum += d * d;
                    }
                }
            });
        });
        if (a.mode == 1) {
            lsum = wave_sum(lsum, 64);
            if (lane == 0) red[wv] = lsum;
            __syncthreads();
            if (tid == 0) a.part[t] = (red[0] + red[1]) + (red[2] + red[3]);
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores + trailing requests drain before the VM counter is counted again
    }
}

// row-major [rows][K] (16-bit) -> k-tiled [K/32][rows][32]. One block: 64 rows x two 32-k slices (whole 128-B lines in,
// 16 rows x 64 B = whole KiB runs out), thread = one 16-B chunk of each slice.
__global__ __launch_bounds__(256) void k_ktile_pack(const char* __restrict__ src, int64_t rows, int64_t K,
                                                    char* __restrict__ dst) {
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    const int64_t row = (int64_t)blockIdx.x * 64 + r;
    if (row >= rows) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t kt = (int64_t)blockIdx.y * 2 + j;
        if (kt * 32 >= K) break;
        const uint4 v = *reinterpret_cast<const uint4*>(src + row * K * 2 + kt * 64 + c * 16);
        *reinterpret_cast<uint4*>(dst + (kt * rows + row) * 64 + c * 16) = v;
    }
}

// sum of the per-tile partials in index order (one workgroup; fixed tree) -> *loss_sum += total
__global__ __launch_bounds__(1024) void k_loss_reduce(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = *out + (float)red[0];
}

// split-K epilogue: Y[i] = round(sum_s part[s][i] (+ bias[i % R])), slices in ascending order; thread = 8 consecutive outputs
template <int DT>
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part, int SK, int64_t NR, int64_t R,
                                                       const char* __restrict__ bias, char* __restrict__ Y) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= NR) return;
    float v[8];
    {
        const float4 a0 = *reinterpret_cast<const float4*>(part + i0), a1 = *reinterpret_cast<const float4*>(part + i0 + 4);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    }
    for (int s = 1; s < SK; ++s) {
        const float* p = part + (int64_t)s * NR + i0;
        const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    }
    uint16_t o[8];
    const int64_t c0 = i0 % R;      // R % 8 == 0: the 8 outputs lie in one row
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float y = v[j] + (bias ? load_as_f32(bias, c0 + j, DT) : 0.0f);
        o[j] = DT == LLMC_BF16 ? f32_to_bf16_bits(y) : f32_to_f16_bits(y);
    }
    uint4 ov;
    __builtin_memcpy(&ov, o, 16);
    *reinterpret_cast<uint4*>(Y + i0 * 2) = ov;
}

// How many k-slices a mode-0 product of this shape is cut into so that tiles x slices fills the persistent grid (an
// evaluation forward of 2048 tokens through a 4096-wide layer is 128 tiles of 256 x 256 for 256 CUs): 1 = no split.
static int lin_ksplit(int64_t N, int64_t K, int64_t R, int grid) {
    const int64_t tiles = ceil_div64(N, LT) * ceil_div64(R, LT);
    if (tiles * 4 >= (int64_t)grid * 3 || R % 8 != 0) return 1;       // >= 3/4 of the CUs busy already
    int sk = (int)(grid / tiles);
    const int by_k = (int)(K / 512);                                   // at least 4 ring turns per slice
    if (sk > by_k) sk = by_k;
    if (sk > 8) sk = 8;
    return sk < 2 ? 1 : sk;
}

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_linear_eval_ws_bytes(int64_t N, int64_t K, int64_t R) {
    if (N <= 0 || R <= 0) return 0;
    const size_t loss = (size_t)(ceil_div64(N, LT) * ceil_div64(R, LT)) * sizeof(float);
    const int sk = K > 0 && K % (L4_KS * L4_RING) == 0 ? lin_ksplit(N, K, R, device_cu_count() & ~7) : 1;   // mode 0 of llmc_linear_eval_kt: fp32 slices
    const size_t split = sk > 1 ? (size_t)sk * N * R * sizeof(float) : 0;
    return loss > split ? loss : split;
}

extern "C" int llmc_linear_eval(const void* X, const void* Wq, int dt, int64_t N, int64_t K, int64_t R, int mode,
                                void* Yout, const void* Y0, float* loss_sum, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "linear_eval: dtype must be f16 or bf16");
    LLMC_REQUIRE(X && Wq && N > 0 && K > 0 && R > 0, "linear_eval: null/empty argument");
    LLMC_REQUIRE((mode == 0 && Yout) || (mode == 1 && Y0 && loss_sum && ws), "linear_eval: outputs for the mode missing");
    LLMC_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Wq & 15) == 0, "linear_eval: operands must be 16-B aligned");
    if (K % LBK != 0 || N * K * 2 >= (1ll << 32) || R * K * 2 >= (1ll << 32)) {
        set_last_error_msg("linear_eval: needs K % 64 == 0 and operands below 4 GiB");
        return LLMC_ENOTSUP;
    }
    hipStream_t st = (hipStream_t)stream;
    LinArgs a;
    a.X = (const char*)X; a.W = (const char*)Wq; a.N = N; a.K = K; a.R = R; a.mode = mode;
    a.Y = (char*)Yout; a.Y0 = (const char*)Y0; a.part = (float*)ws;
    a.ntm = (int)ceil_div64(N, LT); a.ntn = (int)ceil_div64(R, LT);
    lin_tile_order(a, 256);
    a.y0_lds = 0;
    a.yblk = 0;
    a.C = nullptr; a.ldc = 0; a.csign = 1.0f; a.krange = 0; a.kunit = 1; a.queue = nullptr; a.ksplit = 1; a.ntm_real = a.ntm;
    if (dt == LLMC_BF16) {
        if (int rc = ensure_dynamic_lds((const void*)k_linear_eval<LLMC_BF16>, LLDS + 64)) return rc;
        hipLaunchKernelGGL((k_linear_eval<LLMC_BF16>), dim3(256), dim3(LTHREADS), LLDS + 64, st, a);
    } else {
        if (int rc = ensure_dynamic_lds((const void*)k_linear_eval<LLMC_F16>, LLDS + 64)) return rc;
        hipLaunchKernelGGL((k_linear_eval<LLMC_F16>), dim3(256), dim3(LTHREADS), LLDS + 64, st, a);
    }
    LLMC_LAUNCH_CHECK();
    if (mode == 1) {
        hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, st, (const float*)ws, a.ntm * a.ntn, loss_sum);
        LLMC_LAUNCH_CHECK();
    }
    return LLMC_OK;
}

extern "C" int llmc_ktile_pack(const void* src, int dt, int64_t rows, int64_t K, void* dst, llmc_stream_t stream) {
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "ktile_pack: dtype must be f16 or bf16");
    LLMC_REQUIRE(src && dst && src != dst && rows > 0 && K > 0, "ktile_pack: null/empty/in-place argument");
    LLMC_REQUIRE(K % L4_KS == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
                 "ktile_pack: needs K % 32 == 0 and 16-B aligned buffers");
    LLMC_REQUIRE(K / 64 < 65535, "ktile_pack: K too large");
    hipLaunchKernelGGL(k_ktile_pack, dim3((unsigned)ceil_div64(rows, 64), (unsigned)ceil_div64(K, 64)), dim3(256), 0,
                       (hipStream_t)stream, (const char*)src, rows, K, (char*)dst);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" size_t llmc_linear_eval_yblocked_bytes(int64_t N, int64_t R) {
    if (N <= 0 || R <= 0) return 0;
    return (size_t)(ceil_div64(N, LT) * ceil_div64(R, LT)) * (size_t)(LT * LT * 2);
}

extern "C" int llmc_linear_eval_kt(const void* Xt, const void* Wt, int dt, int64_t N, int64_t K, int64_t R, int mode_flags,
                                   void* Yout, const void* Y0, float* loss_sum, void* ws, llmc_stream_t stream) {
    const int mode = mode_flags & 3, yblk = (mode_flags & LLMC_LINEAR_YBLOCKED) != 0;
    LLMC_REQUIRE(dt == LLMC_F16 || dt == LLMC_BF16, "linear_eval_kt: dtype must be f16 or bf16");
    LLMC_REQUIRE(Xt && Wt && N > 0 && K > 0 && R > 0, "linear_eval_kt: null/empty argument");
    LLMC_REQUIRE((mode_flags & ~(1 | LLMC_LINEAR_YBLOCKED)) == 0, "linear_eval_kt: unknown mode bits");
    LLMC_REQUIRE((mode == 0 && Yout) || (mode == 1 && Y0 && loss_sum && ws), "linear_eval_kt: outputs for the mode missing");
    LLMC_REQUIRE(((uintptr_t)Xt & 15) == 0 && ((uintptr_t)Wt & 15) == 0, "linear_eval_kt: operands must be 16-B aligned");
    if (K % (L4_KS * L4_RING) != 0 || N * K * 2 >= (1ll << 32) || R * K * 2 >= (1ll << 32)) {
        set_last_error_msg("linear_eval_kt: needs K % 128 == 0 and operands below 4 GiB");
        return LLMC_ENOTSUP;
    }
    if (yblk) {
        LLMC_REQUIRE((((uintptr_t)(mode == 0 ? (const void*)Yout : Y0)) & 15) == 0, "linear_eval_kt: blocked Y must be 16-B aligned");
        if (llmc_linear_eval_yblocked_bytes(N, R) >= (1ull << 32)) {
            set_last_error_msg("linear_eval_kt: tile-blocked Y must stay below 4 GiB");
            return LLMC_ENOTSUP;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    LinArgs a;
    a.X = (const char*)Xt; a.W = (const char*)Wt; a.N = N; a.K = K; a.R = R; a.mode = mode;
    a.Y = (char*)Yout; a.Y0 = (const char*)Y0; a.part = (float*)ws;
    a.ntm = (int)ceil_div64(N, LT); a.ntn = (int)ceil_div64(R, LT);
    const int grid = device_cu_count() & ~7;
    LLMC_REQUIRE(grid >= 8, "linear_eval_kt: device has fewer than 8 compute units");
    a.yblk = yblk;
    a.C = nullptr; a.ldc = 0; a.csign = 1.0f; a.krange = 0; a.kunit = 1; a.queue = nullptr; a.ksplit = 1; a.ntm_real = a.ntm;
    // split-K for products that leave CUs idle (mode 0, row-major output, workspace given): fp32 slices into ws, then one
    // reduction + rounding pass. option linear_nosplit keeps the single-pass form (tests compare the two).
    const int sk = (mode == 0 && !yblk && ws && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)Yout & 15) == 0 && !opt(OPT_LINEAR_NOSPLIT))
                       ? lin_ksplit(N, K, R, grid) : 1;
    if (sk > 1) {
        a.mode = 3; a.C = (float*)ws; a.ldc = R; a.ksplit = sk; a.ntm = a.ntm_real * sk;
    }
    lin_tile_order(a, grid);
    a.y0_lds = mode == 1 && !yblk && R % 8 == 0 && ((uintptr_t)Y0 & 15) == 0 && N * R * 2 < (1ll << 32);
    int stage = mode == 1 ? (yblk ? 2 : a.y0_lds ? 1 : 0) : 0;
#ifdef LLMC_LAB   // lab builds only (tools/probes): wrong losses by design, never compiled into the shipped library
    if (const char* e = lab_env("LLMC_LIN_ABL")) {
        const int v = atoi(e);
        if (v == 1) { a.mode = 2; a.y0_lds = 0; stage = 0; }      // main loop only
        if (v == 2 && stage == 1) { a.y0_lds = 0; stage = 0; }    // row-major Y0 straight from global
        if (v == 3 && stage) a.y0_lds = 2;                        // Y0 staged and waited for, not folded
    }
#endif
    const bool bf = dt == LLMC_BF16;
    const void* fn = stage == 2 ? (bf ? (const void*)k_linear_eval4<LLMC_BF16, 2> : (const void*)k_linear_eval4<LLMC_F16, 2>)
                   : stage == 1 ? (bf ? (const void*)k_linear_eval4<LLMC_BF16, 1> : (const void*)k_linear_eval4<LLMC_F16, 1>)
                                : (bf ? (const void*)k_linear_eval4<LLMC_BF16, 0> : (const void*)k_linear_eval4<LLMC_F16, 0>);
    if (int rc = ensure_dynamic_lds(fn, L4_LDS + 64)) return rc;
    void* kargs[] = {(void*)&a};
    LLMC_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(L4_THREADS), kargs, (size_t)(L4_LDS + 64), st));
    if (sk > 1) {
        const int64_t NR = N * R;
        const unsigned blocks = (unsigned)ceil_div64(NR, 256 * 8);
        if (bf) hipLaunchKernelGGL((k_splitk_reduce<LLMC_BF16>), dim3(blocks), dim3(256), 0, st, (const float*)ws, sk, NR, R, (const char*)Y0, (char*)Yout);
        else hipLaunchKernelGGL((k_splitk_reduce<LLMC_F16>), dim3(blocks), dim3(256), 0, st, (const float*)ws, sk, NR, R, (const char*)Y0, (char*)Yout);
        LLMC_LAUNCH_CHECK();
    }
    if (mode == 1) {
        hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, st, (const float*)ws, a.ntm * a.ntn, loss_sum);
        LLMC_LAUNCH_CHECK();
    }
    return LLMC_OK;
}

// ---- gemm6: K3's large products C = +-op(A) B on k_linear_eval4 (declared in sgemm.h) --------------------------------
// fp32 operands are split exactly into three bf16 terms (gemm3.hip) ONCE, into k-tiled planes stacked along k: per 32-k
// block six slices, A side (lo, hi, mid, mid, hi, hi), B side (hi, lo, mid, hi, mid, hi), so that the k-sum of the stacked
// product is lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi = the fp32 product to fp32 accuracy (small terms first per
// block). k_gemm3 re-splits every operand element in every tile that uses it and reads a fragment from LDS per MFMA.
namespace llmc {

__device__ __forceinline__ void split3(float v, uint16_t (&pl)[3]) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const uint16_t b = f32_to_bf16_bits(v);
        pl[t] = b;
        v = v - bf16_bits_to_f32(b);
    }
}
// plane (0 hi, 1 mid, 2 lo) of stacked slice q on the A / B side; compile-time so that the unrolled loops index registers
__device__ __forceinline__ constexpr int g6_plane(int side, int q) {
    return side ? (q == 1 ? 2 : (q == 2 || q == 4) ? 1 : 0) : (q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0);
}

// source [rows, Kd] row-major (k contiguous, row stride ld) -> planes T[(Kd/32) * 6][rows][32]. One workgroup: 64 rows x one
// 32-k block; thread = 8 consecutive k of one row (4 threads per row: whole 128-B lines in, 16 rows x 64 B = KiB runs out).
// side 0 = A order, 1 = B order.
__global__ __launch_bounds__(256) void k_split6_rows(const float* __restrict__ src, int64_t ld, int rows, int Kd, int side,
                                                     uint16_t* __restrict__ dst) {
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    const int64_t row = (int64_t)blockIdx.x * 64 + r;
    const int kb = blockIdx.y;
    if (row >= rows) return;
    const float* p = src + row * ld + kb * 32 + c * 8;
    const float4 f0 = *reinterpret_cast<const float4*>(p);
    const float4 f1 = *reinterpret_cast<const float4*>(p + 4);
    const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    uint16_t pl[8][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(v[e], pl[e]);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = side ? pl[e][g6_plane(1, q)] : pl[e][g6_plane(0, q)];
        uint4 ov;
        __builtin_memcpy(&ov, o, 16);
        *reinterpret_cast<uint4*>(dst + (((int64_t)kb * 6 + q) * rows + row) * 32 + c * 8) = ov;
    }
}

// source [Kd, cols] k-major (row stride ld) -> planes T[(Kd/32) * 6][cols][32]: a 32 x 32 tile through LDS.
__global__ __launch_bounds__(256) void k_split6_cols(const float* __restrict__ src, int64_t ld, int cols, int Kd, int side,
                                                     uint16_t* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int kb = blockIdx.y, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int y = ty; y < 32; y += 8) {
        const int c = c0 + tx;
        tile[y][tx] = c < cols ? src[(int64_t)(kb * 32 + y) * ld + c] : 0.0f;
    }
    __syncthreads();
    // thread -> column (row of the planes) c0 + (threadIdx >> 3), 4 consecutive k: (threadIdx & 7) * 4
    const int cr = threadIdx.x >> 3, k4 = (threadIdx.x & 7) * 4;
    if (c0 + cr >= cols) return;
    uint16_t pl[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(tile[k4 + e][cr], pl[e]);
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = side ? pl[e][g6_plane(1, p)] : pl[e][g6_plane(0, p)];
        uint2 ov;
        __builtin_memcpy(&ov, o, 8);
        *reinterpret_cast<uint2*>(dst + (((int64_t)kb * 6 + p) * cols + c0 + cr) * 32 + k4) = ov;
    }
}

static inline size_t g6_align(size_t x) { return (x + 255) & ~(size_t)255; }
size_t gemm6_ws_bytes(int M, int N, int Kd) {
    return g6_align((size_t)6 * Kd * M * 2) + g6_align((size_t)6 * Kd * N * 2) + 512;
}
// C [M, N] (row stride ldc) = sign * A B with A [M, Kd] row-major fp32 (lda), B [Kd, N] k-major fp32 (ldb).
// a_upper: A[i][k] == 0 for k < i; b_upper: B[k][j] == 0 for k > j (at most one of them). Kd % 256 == 0.
int gemm6_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int Kd,
                 int a_upper, int b_upper, float sign, void* ws, hipStream_t st) {
    LLMC_REQUIRE(A && B && C && ws && M > 0 && N > 0 && Kd > 0 && Kd % 256 == 0 && !(a_upper && b_upper), "gemm6: bad argument");
    LLMC_REQUIRE(lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)ws & 255) == 0, "gemm6: alignment");
    LLMC_REQUIRE((int64_t)6 * Kd * M * 2 < (1ll << 32) && (int64_t)6 * Kd * N * 2 < (1ll << 32), "gemm6: planes must stay below 4 GiB");
    uint16_t* Ap = (uint16_t*)ws;
    uint16_t* Bp = (uint16_t*)((char*)ws + g6_align((size_t)6 * Kd * M * 2));
    hipLaunchKernelGGL(k_split6_rows, dim3((M + 63) / 64, Kd / 32), dim3(256), 0, st, A, lda, M, Kd, 0, Ap);
    LLMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_split6_cols, dim3((N + 31) / 32, Kd / 32), dim3(256), 0, st, B, ldb, N, Kd, 1, Bp);
    LLMC_LAUNCH_CHECK();
    LinArgs a;
    a.X = (const char*)Ap; a.W = (const char*)Bp; a.N = M; a.K = (int64_t)6 * Kd; a.R = N; a.mode = 3;
    a.Y = nullptr; a.Y0 = nullptr; a.part = nullptr;
    a.ntm = (int)ceil_div64(M, LT); a.ntn = (int)ceil_div64(N, LT);
    const int grid = device_cu_count() & ~7;
    LLMC_REQUIRE(grid >= 8, "gemm6: device has fewer than 8 compute units");
    lin_tile_order(a, grid);
    a.yblk = 0; a.y0_lds = 0;
    a.C = C; a.ldc = ldc; a.csign = sign; a.krange = a_upper ? 1 : b_upper ? 2 : 0; a.kunit = 6;
    a.queue = (unsigned*)((char*)Bp + g6_align((size_t)6 * Kd * N * 2));
    a.ksplit = 1; a.ntm_real = a.ntm;
    LLMC_HIP_CHECK(hipMemsetAsync(a.queue, 0, 4, st));
    const void* fn = (const void*)k_linear_eval4<LLMC_BF16, 0>;
    if (int rc = ensure_dynamic_lds(fn, L4_LDS + 64)) return rc;
    void* kargs[] = {(void*)&a};
    LLMC_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(L4_THREADS), kargs, (size_t)(L4_LDS + 64), st));
    return LLMC_OK;
}
}  // namespace llmc
