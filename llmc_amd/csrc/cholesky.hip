// cholesky.hip — K2/K3 of the GPTQ path (gptq.py:139-174):
//   llmc_hessian_prep     dead-column fix, actorder gather of H and W, damping
//   llmc_chol_inv_upper   U upper with H^-1 = U^T U, the factor the reference obtains from
//                         cholesky -> cholesky_inverse -> cholesky(upper)
//
// Math (fp32 throughout, like the reference): with J the index reversal, U = J * inv(chol_lower(J H J)) * J.
// In upper/row-major form: A' = antitranspose(H); A' = U'^T U' (blocked right-looking, 128-wide panels:
// one workgroup factors + inverts the diagonal block, the panel solve and the symmetric trailing update
// are fp32-MFMA GEMMs); V = U'^-1 by recursive doubling over the already inverted diagonal blocks
// ([[A,C],[0,B]]^-1 = [[A^-1, -A^-1 C B^-1],[0, B^-1]]); U = antitranspose(V).  2K^3/3 flops instead of
// the reference's 4K^3/3.
#include "common.h"
#include "sgemm.h"

namespace llmc {

static constexpr int NB = 128;

// ---------------------------------------------------------------------------------------------
// out[i][j] = in[n-1-j][n-1-i]  (reflection across the anti-diagonal), 32x32 LDS tiles, coalesced both
// ways. upper_only: write 0 for j < i.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_antitranspose(const float* __restrict__ in, float* __restrict__ out,
                                                       int n, int upper_only) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;    // input tile
    for (int y = ty; y < 32; y += 8) {
        int r = r0 + y, c = c0 + tx;
        tile[y][tx] = (r < n && c < n) ? in[(int64_t)r * n + c] : 0.0f;
    }
    __syncthreads();
    // in[r][c] -> out[n-1-c][n-1-r]; output row = n-1-(c0+y'), output col = n-1-(r0+x')
    for (int y = ty; y < 32; y += 8) {
        int c = c0 + y, r = r0 + tx;
        if (r < n && c < n) {
            int orow = n - 1 - c, ocol = n - 1 - r;
            float v = tile[tx][y];
            if (upper_only && ocol < orow) v = 0.0f;
            out[(int64_t)orow * n + ocol] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Diagonal block: U = chol_upper(D) in LDS, V = U^-1. One workgroup of 512 threads.
//   W   : work matrix (ld), diagonal block at (k0,k0), size nb <= 128. U (upper, strict lower zeroed) is
//         written back; Vout [128 x 128] (ld 128) receives U^-1 (zero padded).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_potrf_inv(float* __restrict__ W, int64_t ld, int k0, int nb,
                                                   float* __restrict__ Vout, int* __restrict__ info) {
    __shared__ float S[NB][NB + 1];
    __shared__ float V[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 512) {
        int i = e >> 7, j = e & 127;
        float v = 0.0f;
        if (i < nb && j < nb && j >= i) v = W[(int64_t)(k0 + i) * ld + k0 + j];
        S[i][j] = v;
        V[i][j] = 0.0f;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        float d = S[j][j];
        if (!(d > 0.0f) && tid == 0) atomicCAS(info, 0, k0 + j + 1);
        float rj = sqrtf(d);
        __syncthreads();
        if (tid == 0) S[j][j] = rj;
        for (int l = j + 1 + tid; l < nb; l += 512) S[j][l] = S[j][l] / rj;
        __syncthreads();
        // trailing rank-1 update of the upper triangle: rows i in (j, nb), cols l >= i
        const int m = nb - j - 1;
        for (int e = tid; e < m * m; e += 512) {
            int i = j + 1 + e / m, l = j + 1 + e % m;
            if (l >= i) S[i][l] -= S[j][i] * S[j][l];
        }
        __syncthreads();
    }
    // V = U^-1 by back substitution; thread j owns column j.
    if (tid < nb) {
        const int j = tid;
        for (int i = j; i >= 0; --i) {
            float s = (i == j) ? 1.0f : 0.0f;
            for (int k = i + 1; k <= j; ++k) s -= S[i][k] * V[k][j];
            V[i][j] = s / S[i][i];
        }
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 512) {
        int i = e >> 7, j = e & 127;
        Vout[e] = V[i][j];
        if (i < nb && j < nb) W[(int64_t)(k0 + i) * ld + k0 + j] = (j >= i) ? S[i][j] : 0.0f;
    }
}

// copy the inverted diagonal blocks into the work matrix (upper), before the doubling levels
__global__ __launch_bounds__(256) void k_place_diag_inv(float* __restrict__ W, int64_t ld, int K,
                                                        const float* __restrict__ Vbuf) {
    const int b = blockIdx.x;
    const int k0 = b * NB;
    const int nb = min(NB, K - k0);
    const float* V = Vbuf + (int64_t)b * NB * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 7, j = e & 127;
        if (i < nb && j < nb) W[(int64_t)(k0 + i) * ld + k0 + j] = V[e];
    }
}

// ---------------------------------------------------------------------------------------------
// K2 helpers
// ---------------------------------------------------------------------------------------------
// pass 1: dead fix on the diagonal, dead flags, sum of the fixed diagonal (fixed-order tree: deterministic)
__global__ __launch_bounds__(1024) void k_diag_fix(float* __restrict__ H, int K, uint8_t* __restrict__ dead,
                                                   float* __restrict__ diag_mean) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < K; i += 1024) {
        float d = H[(int64_t)i * K + i];
        uint8_t dd = (d == 0.0f);
        if (dd) {
            d = 1.0f;
            H[(int64_t)i * K + i] = 1.0f;
        }
        dead[i] = dd;
        s += (double)d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *diag_mean = (float)(red[0] / (double)K);
}

// Hout[i][j] = H[perm[i]][perm[j]] + (i == j) * percdamp * mean(diag)
__global__ __launch_bounds__(256) void k_gather_h(const float* __restrict__ H, int K,
                                                  const int64_t* __restrict__ perm, float percdamp,
                                                  const float* __restrict__ diag_mean, float* __restrict__ Hout) {
    const int i = blockIdx.y;
    const int64_t pi = perm ? perm[i] : i;
    const float damp = percdamp * (*diag_mean);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < K; j += gridDim.x * 256) {
        const int64_t pj = perm ? perm[j] : j;
        float v = H[pi * K + pj];
        if (i == j) v += damp;
        Hout[(int64_t)i * K + j] = v;
    }
}

// Wout[r][j] = dead[perm[j]] ? 0 : float(W[r][perm[j]])
template <typename T>
__global__ __launch_bounds__(256) void k_gather_w(const T* __restrict__ W, int64_t R, int K,
                                                  const int64_t* __restrict__ perm,
                                                  const uint8_t* __restrict__ dead, float* __restrict__ Wout) {
    const int64_t r = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < K; j += gridDim.x * 256) {
        const int64_t pj = perm ? perm[j] : j;
        float v = dead[pj] ? 0.0f : to_f32<T>(W[r * K + pj]);
        Wout[r * K + j] = v;
    }
}

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_hessian_prep_ws_bytes(int64_t K) {
    if (K <= 0) return 0;
    return (size_t)(((K + 63) / 64) * 64 + 64);
}

extern "C" int llmc_hessian_prep(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm,
                                 float percdamp, float* Hout, float* Wout, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(H && ws && K > 0, "hessian_prep: null/empty argument");
    LLMC_REQUIRE((W && Wout && R > 0) || (!W && !Wout), "hessian_prep: W and Wout go together");
    LLMC_REQUIRE(dtype_ok(wdt), "hessian_prep: bad dtype");
    LLMC_REQUIRE(Hout != H, "hessian_prep: Hout must not alias H");
    LLMC_REQUIRE(K < (1ll << 31) && R < 65536ll * 32768ll, "hessian_prep: shape too large");
    hipStream_t st = (hipStream_t)stream;
    float* diag_mean = (float*)ws;
    uint8_t* dead = (uint8_t*)ws + 64;
    hipLaunchKernelGGL(k_diag_fix, dim3(1), dim3(1024), 0, st, H, (int)K, dead, diag_mean);
    LLMC_LAUNCH_CHECK();
    int gx = (int)ceil_div64(K, 256 * 4);
    if (Hout) {
        hipLaunchKernelGGL(k_gather_h, dim3(gx, (unsigned)K), dim3(256), 0, st, (const float*)H, (int)K, perm,
                           percdamp, (const float*)diag_mean, Hout);
        LLMC_LAUNCH_CHECK();
    }
    // grid.y is limited to 65535: loop over row slabs
    for (int64_t r0 = 0; W && r0 < R; r0 += 32768) {
        int64_t rows = R - r0 < 32768 ? R - r0 : 32768;
        dim3 grid(gx, (unsigned)rows);
        if (wdt == LLMC_F16)
            hipLaunchKernelGGL((k_gather_w<f16_t>), grid, dim3(256), 0, st, (const f16_t*)W + r0 * K, rows, (int)K,
                               perm, (const uint8_t*)dead, Wout + r0 * K);
        else if (wdt == LLMC_BF16)
            hipLaunchKernelGGL((k_gather_w<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)W + r0 * K, rows, (int)K,
                               perm, (const uint8_t*)dead, Wout + r0 * K);
        else
            hipLaunchKernelGGL((k_gather_w<float>), grid, dim3(256), 0, st, (const float*)W + r0 * K, rows, (int)K,
                               perm, (const uint8_t*)dead, Wout + r0 * K);
        LLMC_LAUNCH_CHECK();
    }
    return LLMC_OK;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t llmc_chol_inv_upper_ws_bytes(int64_t K) {
    if (K <= 0) return 0;
    size_t work = align256((size_t)K * K * 4);
    size_t vbuf = align256((size_t)ceil_div64(K, NB) * NB * NB * 4);
    size_t xbuf = align256((size_t)(K / 2 + NB) * (K / 2 + NB) * 4);
    return work + vbuf + xbuf;
}

extern "C" int llmc_chol_inv_upper(float* A, int64_t K64, void* ws, int32_t* info_dev, llmc_stream_t stream) {
    LLMC_REQUIRE(A && ws && info_dev && K64 > 0, "chol_inv_upper: null/empty argument");
    LLMC_REQUIRE(K64 % 4 == 0 && K64 < (1 << 30), "chol_inv_upper: K must be a multiple of 4");
    LLMC_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)ws & 255) == 0, "chol_inv_upper: alignment");
    hipStream_t st = (hipStream_t)stream;
    const int K = (int)K64;
    float* Wk = (float*)ws;
    float* Vbuf = (float*)((char*)ws + align256((size_t)K * K * 4));
    float* Xbuf = (float*)((char*)Vbuf + align256((size_t)ceil_div64(K, NB) * NB * NB * 4));
    LLMC_HIP_CHECK(hipMemsetAsync(info_dev, 0, 4, st));

    dim3 tgrid((K + 31) / 32, (K + 31) / 32);
    hipLaunchKernelGGL(k_antitranspose, tgrid, dim3(256), 0, st, (const float*)A, Wk, K, 0);
    LLMC_LAUNCH_CHECK();

    const int nblk = (K + NB - 1) / NB;
    // ---- blocked upper Cholesky Wk = U'^T U'
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * NB;
        const int nb = K - k0 < NB ? K - k0 : NB;
        float* Vb = Vbuf + (size_t)b * NB * NB;
        hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(512), 0, st, Wk, (int64_t)K, k0, nb, Vb, info_dev);
        LLMC_LAUNCH_CHECK();
        const int nrem = K - k0 - nb;
        if (nrem <= 0) break;
        float* P = Wk + (size_t)k0 * K + k0 + nb;  // panel rows k0..k0+nb, cols k0+nb..K
        SgemmArgs g{};
        // panel solve: P = V^T P  (op(A)[i][k] = V[k][i], lower triangular)
        g.A = Vb; g.lda = NB; g.B = P; g.ldb = K; g.C = P; g.ldc = K;
        g.M = g.M_last = nb; g.N = g.N_last = nrem; g.Kd = g.Kd_last = nb;
        g.epilogue = SG_SET; g.a_lower = 1; g.batch = 1;
        int rc = sgemm_launch(g, true, false, st);
        if (rc) return rc;
        // trailing update: T -= P^T P on the upper tiles
        SgemmArgs u{};
        u.A = P; u.lda = K; u.B = P; u.ldb = K;
        u.C = Wk + (size_t)(k0 + nb) * K + k0 + nb; u.ldc = K;
        u.M = u.M_last = nrem; u.N = u.N_last = nrem; u.Kd = u.Kd_last = nb;
        u.epilogue = SG_SUB; u.c_upper_only = 1; u.batch = 1;
        rc = sgemm_launch(u, true, false, st);
        if (rc) return rc;
    }
    // ---- V = U'^-1: inverted diagonal blocks, then doubling levels
    hipLaunchKernelGGL(k_place_diag_inv, dim3(nblk), dim3(256), 0, st, Wk, (int64_t)K, K, (const float*)Vbuf);
    LLMC_LAUNCH_CHECK();
    for (int64_t h = NB; h < K; h *= 2) {
        const int npairs = (int)((K - h + 2 * h - 1) / (2 * h));  // pairs with a non-empty right block
        if (npairs <= 0) break;
        const int64_t o_last = (int64_t)(npairs - 1) * 2 * h;
        const int n2_last = (int)((K - o_last - h) < h ? (K - o_last - h) : h);
        const int64_t stride = 2 * h * ((int64_t)K + 1);
        // X = A^-1 C
        SgemmArgs x{};
        x.A = Wk; x.lda = K; x.sA = stride;                 // A^-1 at (o, o), upper
        x.B = Wk + h; x.ldb = K; x.sB = stride;             // C at (o, o+h)
        // X is [h x n2]: with one pair its leading dimension shrinks to n2 (keeps X within K^2/4 floats)
        const int64_t ldX = npairs == 1 ? ((n2_last + 3) / 4) * 4 : h;
        x.C = Xbuf; x.ldc = ldX; x.sC = h * h;
        x.M = x.M_last = (int)h; x.N = (int)h; x.N_last = n2_last; x.Kd = x.Kd_last = (int)h;
        x.epilogue = SG_SET; x.a_upper = 1; x.batch = npairs;
        int rc = sgemm_launch(x, false, false, st);
        if (rc) return rc;
        // C = -X B^-1
        SgemmArgs y{};
        y.A = Xbuf; y.lda = ldX; y.sA = h * h;
        y.B = Wk + h * ((int64_t)K + 1); y.ldb = K; y.sB = stride;   // B^-1 at (o+h, o+h), upper
        y.C = Wk + h; y.ldc = K; y.sC = stride;
        y.M = y.M_last = (int)h; y.N = (int)h; y.N_last = n2_last; y.Kd = (int)h; y.Kd_last = n2_last;
        y.epilogue = SG_NEG; y.b_upper = 1; y.batch = npairs;
        rc = sgemm_launch(y, false, false, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_antitranspose, tgrid, dim3(256), 0, st, (const float*)Wk, A, K, 1);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
