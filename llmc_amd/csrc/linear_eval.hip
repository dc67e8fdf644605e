// linear_eval.hip — K9: the fake-quant W4A16 matmul of AWQ's scale search (awq.py:110-145, 229-236):
//   Y = X [N, K] . Wq [R, K]^T   16-bit operands, fp32 accumulation on v_mfma_f32_32x32x16, Y rounded to the
//   model dtype exactly once (what F.linear returns), then either stored (get_original_out) or compared
//   with the stored original output: loss = sum((Y0 - Y)^2), the difference formed in the model dtype like
//   the reference's `(org_out - out).float().pow(2)`.
// Both operands are contiguous along the contraction axis ("NT"): tiles are staged row-major by LDS-DMA and
// read with ds_read_b128; the 16-B chunk index is XOR-ed with ((row >> 1) & 7) on the DMA source address
// and on the read, which makes every ds_read_b128 lane group hit 16 distinct 16-B bank slots.
// Tile 256 x 256, K-step 64, 8 waves (2 x 4), persistent grid with XCD-contiguous tile ranges: the 32
// workgroups of an XCD walk 2 token panels x 16 weight panels at the same K position (L2-resident).
#include "common.h"
#include "mfma_common.h"

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
};

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

    const int G = gridDim.x;
    const int lw = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int ntiles = a.ntm * a.ntn;
    const int nk = (int)(a.K / LBK);

    for (int t = lw; t < ntiles; t += G) {
        const int tm = t / a.ntn, tn = t - tm * a.ntn;
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

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_linear_eval_ws_bytes(int64_t N, int64_t K, int64_t R) {
    if (N <= 0 || R <= 0) return 0;
    return (size_t)(ceil_div64(N, LT) * ceil_div64(R, LT)) * sizeof(float);
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
