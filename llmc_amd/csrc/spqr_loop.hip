// spqr_loop.hip — SpQR.weight_transform (spqr.py:185-254): the blocked column loop with leave-one-out outlier detection
// and the round_zp=False quantizer, as the reference executes it on asymmetric per-group weights.
//
// Same ownership as gptq_loop.hip: a wave owns 4 rows, 16 lanes per row, lane p owns columns p, p+16, ..., p+112 of the
// 128-column block, so the 16 columns of a detection stripe sit in the 16 lanes of the row (one register index) and a
// group of g = 16 M columns is M registers per lane. Per group start (spqr.py:214-229):
//   pass A  min / second min / multiplicity (and max) of the group by 16-lane broadcasts -> min / max WITHOUT column j
//           is the second extreme iff j holds the only copy of the first (spqr.py:187-189: LooG = G without column j);
//   pass B  Base = sum_k ((qdq(G_k) - G_k) / d_k)^2 and, per owned column j, Loo_j = the same sum without k = j under
//           the qparams of G-without-j, every sum in ascending k (spqr.py:191-201);
//   pass C  M_j = Base - Loo_j > threshold, mean of the kept columns, G' = G (1 - M) + mean M (spqr.py:221-227);
//   (s, z)  asym round_zp=False qparams of G', then the reference's second-level quantizers, which see [R, 1] tensors,
//           leave them ungrouped and return fl(fl(v / ss) * ss), ss = 1e-5 / (qmax - qmin) (spqr.py:323-345).
// Per column (spqr.py:233-250): q, err = (w - q) / d, mask = err^2 > threshold, a masked weight keeps its value
// (err = (w - w) / d), tmp = w, loss = err^2, w_j -= fl(err * U[i][j]) for the later columns of the block.
// The trailing update W[:, i2:] -= Err1 @ U[i1:i2, i2:] is gptq_loop.hip's: sgemm.hip's k-ordered fp32 fma chain.
// Operation order is oracle/csrc/spqr_canon.c's (pinned bit-exactly against the reference: tests/golden/spqr.npz).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "sgemm.h"

namespace llmc {

static constexpr int SBS = 128;   // blocksize
static constexpr int SNT = 256;   // threads per workgroup: 4 waves x 4 rows

struct SpqrBlockArgs {
    const float* W;       // [R, K] running weights
    const float* U;       // [K, K] upper factor
    float* Wout;          // [R, K] tmp
    float* losses;        // [R, K]
    uint8_t* mask;        // [R, K]
    float* Err;           // [R, 128] err of this block
    float* scales;        // [R, K/g]
    float* zeros;         // [R, K/g]
    int64_t R;
    int K, i1, count, ng;
    float qmin, qmax, threshold;
    int detect;           // leave-one-out detection on (not simplified_outliers, threshold finite)
    int use_mask;         // threshold finite
    float sqmin, sqmax, zqmin, zqmax;
};

template <int I, int N, typename F> __device__ __forceinline__ void sp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sp_static_for<I + 1, N>(f);
    }
}

// lane po of every 16-lane row (ds_bpermute: po is a loop counter, the 16 steps of a stripe are a real loop — unrolled,
// the 128 steps with their IEEE divisions and hoisted LDS reads spill hundreds of registers)
__device__ __forceinline__ float sp_bcast(float v, int po) {
    const int src = ((int)(threadIdx.x & 48) | po) << 2;
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v)));
}

__device__ __forceinline__ void qp_asym_nr(float mn, float mx, float qmin, float qmax, float& s, float& z) {
    float d = mx - mn;
    if (d < 1e-5f) d = 1e-5f;
    s = d / (qmax - qmin);
    z = qmin - (mn / s);
}
__device__ __forceinline__ float qdq_nr(float x, float s, float z, float qmin, float qmax) {
    const float sc = s < 1e-9f ? 1e-9f : s;
    float t = x / sc;
    t = t + z;
    t = rintf(t);
    t = fminf(fmaxf(t, qmin), qmax);
    t = t - z;
    return t * s;
}
__device__ __forceinline__ float second_level(float v, float lqmin, float lqmax) {
    const float ss = 1e-5f / (lqmax - lqmin);
    const float sc = ss < 1e-9f ? 1e-9f : ss;
    const float zs = lqmin - (v / ss);
    float t = v / sc;
    t = t + zs;
    t = rintf(t);
    t = fminf(fmaxf(t, lqmin), lqmax);
    t = t - zs;
    return t * ss;
}

// first / second extreme and multiplicity of the first, fed in any order (min and max are order-free)
struct Ext2 {
    float a, b;
    int n;
};
__device__ __forceinline__ void ext_min(Ext2& e, float x) {
    if (x < e.a) { e.b = e.a; e.a = x; e.n = 1; }
    else if (x == e.a) ++e.n;
    else if (x < e.b) e.b = x;
}
__device__ __forceinline__ void ext_max(Ext2& e, float x) {
    if (x > e.a) { e.b = e.a; e.a = x; e.n = 1; }
    else if (x == e.a) ++e.n;
    else if (x > e.b) e.b = x;
}

// GSZ = group size (16 M); one workgroup = 16 rows
template <int GSZ>
__global__ __launch_bounds__(SNT) void k_spqr_block(SpqrBlockArgs a) {
    constexpr int M = GSZ / 16;
    __shared__ __attribute__((aligned(16))) float Us[SBS * SBS];   // Us[i][p*8 + e] = U[i1+i][i1+p+16e] above the diagonal, else 0
    __shared__ float dg[SBS];
    const int tid = threadIdx.x;
    {
        constexpr int NV = SBS * SBS / 4 / SNT;
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + SNT * j;
            const int i = idx >> 5, c4 = (idx & 31) * 4;
            v[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (i < a.count && c4 < a.count) v[j] = *reinterpret_cast<const float4*>(a.U + (int64_t)(a.i1 + i) * a.K + a.i1 + c4);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + SNT * j;
            const int i = idx >> 5, c4 = (idx & 31) * 4;
            const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = c4 + t;
                if (c == i) dg[i] = i < a.count ? vv[t] : 1.0f;
                Us[i * SBS + (c & 15) * 8 + (c >> 4)] = c > i ? vv[t] : 0.0f;
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int p = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * (SNT / 64) + (tid >> 6)) * 4 + (lane >> 4);
    const bool active = row < a.R;
    const int64_t rr = active ? row : a.R - 1;

    float w[8], er[8], ls[8];
    unsigned mk = 0;                       // bit e: column p + 16 e is an outlier
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = p + 16 * e;
        w[e] = (c < a.count) ? a.W[rr * a.K + a.i1 + c] : 0.0f;
        er[e] = 0.0f;
        ls[e] = 0.0f;
    }
    const float* us = Us + p * 8;
    float s = 1.0f, z = 0.0f;
    const float thr = a.threshold, qmin = a.qmin, qmax = a.qmax;

    sp_static_for<0, 8>([&](auto ec) {
        constexpr int E = decltype(ec)::value;
        if constexpr (E % M == 0) {
            if (16 * E < a.count) {
                // ---- group start: columns 16E .. 16E + GSZ - 1 of the block = registers E .. E+M-1 of the 16 lanes
                float mn, mx;
                if (a.detect) {
                    Ext2 lo{INFINITY, INFINITY, 0}, hi{-INFINITY, -INFINITY, 0};
                    sp_static_for<0, M>([&](auto tc) {
                        constexpr int T = decltype(tc)::value;
#pragma unroll 1
                        for (int po = 0; po < 16; ++po) {
                            const float x = sp_bcast(w[E + T], po);
                            ext_min(lo, x);
                            ext_max(hi, x);
                        }
                    });
                    float bs, bz, lsj[M], lzj[M], loo[M], base = 0.0f;
                    qp_asym_nr(lo.a, hi.a, qmin, qmax, bs, bz);
#pragma unroll
                    for (int t = 0; t < M; ++t) {
                        const float x = w[E + t];
                        const float lmn = (x == lo.a && lo.n == 1) ? lo.b : lo.a;
                        const float lmx = (x == hi.a && hi.n == 1) ? hi.b : hi.a;
                        qp_asym_nr(lmn, lmx, qmin, qmax, lsj[t], lzj[t]);
                        loo[t] = 0.0f;
                    }
                    sp_static_for<0, M>([&](auto tc) {
                        constexpr int T = decltype(tc)::value;
#pragma unroll 1
                        for (int po = 0; po < 16; ++po) {
                            const float x = sp_bcast(w[E + T], po);
                            const float d = dg[16 * (E + T) + po];
                            const float eb = (qdq_nr(x, bs, bz, qmin, qmax) - x) / d;
                            base = base + eb * eb;
#pragma unroll
                            for (int t = 0; t < M; ++t) {
                                const float el = (qdq_nr(x, lsj[t], lzj[t], qmin, qmax) - x) / d;
                                const float add = loo[t] + el * el;
                                loo[t] = (t == T && p == po) ? loo[t] : add;     // k == j is left out
                            }
                        }
                    });
                    float mj[M];
#pragma unroll
                    for (int t = 0; t < M; ++t) mj[t] = (base - loo[t]) > thr ? 1.0f : 0.0f;
                    float sum_keep = 0.0f, n_keep = 0.0f;
                    sp_static_for<0, M>([&](auto tc) {
                        constexpr int T = decltype(tc)::value;
#pragma unroll 1
                        for (int po = 0; po < 16; ++po) {
                            const float x = sp_bcast(w[E + T], po);
                            const float m = sp_bcast(mj[T], po);
                            sum_keep = sum_keep + x * (1.0f - m);
                            n_keep = n_keep + (1.0f - m);
                        }
                    });
                    const float mean = sum_keep / (n_keep < 1.0f ? 1.0f : n_keep);
                    mn = INFINITY;
                    mx = -INFINITY;
#pragma unroll
                    for (int t = 0; t < M; ++t) {
                        const float v = w[E + t] * (1.0f - mj[t]) + mean * mj[t];
                        mn = fminf(mn, v);
                        mx = fmaxf(mx, v);
                    }
                } else {
                    mn = INFINITY;
                    mx = -INFINITY;
#pragma unroll
                    for (int t = 0; t < M; ++t) {
                        mn = fminf(mn, w[E + t]);
                        mx = fmaxf(mx, w[E + t]);
                    }
                }
                mn = wave_min(mn, 16);
                mx = wave_max(mx, 16);
                float s1, z1;
                qp_asym_nr(mn, mx, qmin, qmax, s1, z1);
                s = second_level(s1, a.sqmin, a.sqmax);
                z = second_level(z1, a.zqmin, a.zqmax);
                if (active && p == 0) {
                    const int g = (a.i1 + 16 * E) / GSZ;
                    a.scales[row * a.ng + g] = s;
                    a.zeros[row * a.ng + g] = z;
                }
            }
        }
        // ---- the 16 column steps of this stripe
#pragma unroll 1
        for (int po = 0; po < 16; ++po) {
            const int i = 16 * E + po;
            const float wv = sp_bcast(w[E], po);
            const float d = dg[i];
            const float q = qdq_nr(wv, s, z, qmin, qmax);
            float e1 = (wv - q) / d;
            bool out = false;
            if (a.use_mask) {
                out = (e1 * e1) > thr;
                const float Mf = out ? 1.0f : 0.0f;
                const float newq = q * (1.0f - Mf) + wv * Mf;
                e1 = (wv - newq) / d;
            }
            if (p == po) {
                er[E] = e1;
                ls[E] = e1 * e1;
                mk |= out ? (1u << E) : 0u;
            }
            const float* ur = us + i * SBS;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = e1 * ur[e];
                w[e] = w[e] - t;
            }
        }
    });

    if (!active) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = p + 16 * e;
        if (c < a.count) {
            const int64_t o = row * a.K + a.i1 + c;
            a.Wout[o] = w[e];
            a.losses[o] = ls[e];
            a.mask[o] = (mk >> e) & 1u;
        }
        a.Err[row * SBS + c] = (c < a.count) ? er[e] : 0.0f;
    }
}

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_spqr_quantize_ws_bytes(int64_t R, int64_t K) {
    if (R <= 0 || K <= 0) return 0;
    return (size_t)R * SBS * sizeof(float);
}

extern "C" int llmc_spqr_quantize(float* W, const float* Hinv, int64_t R, int64_t K, float qmin, float qmax,
                                  int64_t group_size, float threshold, int simplified_outliers, float scale_qmin,
                                  float scale_qmax, float zero_qmin, float zero_qmax, float* scales, float* zeros,
                                  float* Wout, float* losses, uint8_t* mask, int blocksize, void* ws,
                                  llmc_stream_t stream) {
    LLMC_REQUIRE(W && Hinv && Wout && losses && mask && scales && zeros && ws && R > 0 && K > 0,
                 "spqr_quantize: null/empty argument");
    LLMC_REQUIRE(blocksize == SBS, "spqr_quantize: blocksize must be 128");
    LLMC_REQUIRE(K % 4 == 0 && K < (1 << 30), "spqr_quantize: K must be a multiple of 4");
    LLMC_REQUIRE(threshold == threshold && threshold >= 0.0f, "spqr_quantize: threshold must be >= 0 (inf = no outliers)");
    const int gsz = (int)group_size;
    if (!(gsz == 16 || gsz == 32 || gsz == 64 || gsz == 128) || K % gsz != 0) {
        set_last_error_msg("spqr_quantize: group_size must be 16, 32, 64 or 128 and divide K");
        return LLMC_ENOTSUP;
    }
    hipStream_t st = (hipStream_t)stream;
    float* Err = (float*)ws;
    const bool finite = !(threshold > 3.0e38f);
    for (int64_t i1 = 0; i1 < K; i1 += SBS) {
        const int count = (int)(K - i1 < SBS ? K - i1 : SBS);
        SpqrBlockArgs a;
        a.W = W; a.U = Hinv; a.Wout = Wout; a.losses = losses; a.mask = mask; a.Err = Err;
        a.scales = scales; a.zeros = zeros; a.R = R; a.K = (int)K; a.i1 = (int)i1; a.count = count; a.ng = (int)(K / gsz);
        a.qmin = qmin; a.qmax = qmax; a.threshold = finite ? threshold : INFINITY;
        a.detect = finite && !simplified_outliers; a.use_mask = finite;
        a.sqmin = scale_qmin; a.sqmax = scale_qmax; a.zqmin = zero_qmin; a.zqmax = zero_qmax;
        const int grid = (int)ceil_div64(R, SNT / 16);
        switch (gsz) {
            case 16: hipLaunchKernelGGL((k_spqr_block<16>), dim3(grid), dim3(SNT), 0, st, a); break;
            case 32: hipLaunchKernelGGL((k_spqr_block<32>), dim3(grid), dim3(SNT), 0, st, a); break;
            case 64: hipLaunchKernelGGL((k_spqr_block<64>), dim3(grid), dim3(SNT), 0, st, a); break;
            default: hipLaunchKernelGGL((k_spqr_block<128>), dim3(grid), dim3(SNT), 0, st, a); break;
        }
        LLMC_LAUNCH_CHECK();
        const int64_t i2 = i1 + count;
        if (i2 < K) {   // W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]  (spqr.py:254), the exact fp32 chain of sgemm.hip
            SgemmArgs g{};
            g.A = Err; g.lda = SBS;
            g.B = Hinv + i1 * K + i2; g.ldb = K;
            g.C = W + i2; g.ldc = K;
            g.M = g.M_last = (int)R; g.N = g.N_last = (int)(K - i2); g.Kd = g.Kd_last = count;
            g.epilogue = SG_SUB; g.batch = 1;
            int rc = sgemm_launch(g, false, false, st);
            if (rc) return rc;
        }
    }
    return LLMC_OK;
}
