/*
 * gptq_canon.c — C restatement of GPTQ.weight_transform (llmc/compression/quantization/gptq.py:199-244)
 * with the quantizer of search_column_qparams (gptq.py:359-366 -> quant.py:545-559, 699-717) in the
 * reference's exact fp32 operation order. TEST INFRASTRUCTURE (oracle): used by tests/ as the checker and
 * by bench.py as the timed CPU baseline ("port"), never by the product path.
 *
 * Order of operations per weight element, as the reference executes them on CPU:
 *   in block:   w_j <- w_j - fl(e_i * U[i][j])            for i < j, i ascending          (gptq.py:240)
 *   trailing:   w_j <- w_j - chain_k fmaf(E[k], U[k][j])  chain from +0, k ascending      (gptq.py:244;
 *               MKL sgemm with inner dim 128 is this chain bit for bit — pinned by tests/golden/gptq.npz)
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fopenmp (see oracle/Makefile). Rows are independent ->
 * OpenMP over rows (the CPU baseline uses all host cores; core count is reported by bench.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s, z; } qp_t;

static qp_t qparams_f32(float mn, float mx, int sym, float qmin, float qmax) {
    qp_t q;
    if (sym) {
        float a = fmaxf(fabsf(mx), fabsf(mn));
        if (a < 1e-5f) a = 1e-5f;
        q.s = a / qmax;
        q.z = 0.0f;
    } else {
        float d = mx - mn;
        if (d < 1e-5f) d = 1e-5f;
        q.s = d / (qmax - qmin);
        float r = rintf(mn / q.s);
        float z = qmin - r;
        q.z = fminf(fmaxf(z, qmin), qmax);
    }
    return q;
}

static inline float qdq_f32(float x, float s, float z, float qmin, float qmax) {
    float t = rintf(x / s);
    t = t + z;
    t = fminf(fmaxf(t, qmin), qmax);
    t = t - z;
    return t * s;
}

/*
 * W      [R,K] in: permuted/dead-fixed weights; out: running weights (trailing part updated)
 * Hinv   [K,K] upper factor U
 * group_size 0 = per_channel (static qparams, one group); static_groups: scales/zeros are inputs
 * [R, ng] in original column order, col_group[i] selects the group of processed column i.
 * dynamic: scales/zeros [R, ng] outputs in processing order. zeros may be NULL when sym && static.
 * n_quant < K is OWQ (gptq.py:44-56, 199-221): the loop visits the first n_quant columns only, groups are clipped
 * at n_quant, the trailing columns still receive every block's error feedback.
 */
int gptq_weight_transform_cols(float* W, const float* Hinv, int64_t R, int64_t K, int64_t n_quant, int sym, float qmin,
                               float qmax, int64_t group_size, int static_groups, const int32_t* col_group,
                               float* scales, float* zeros, float* Wout, float* losses, int blocksize);

int gptq_weight_transform(float* W, const float* Hinv, int64_t R, int64_t K, int sym, float qmin, float qmax,
                          int64_t group_size, int static_groups, const int32_t* col_group, float* scales,
                          float* zeros, float* Wout, float* losses, int blocksize) {
    return gptq_weight_transform_cols(W, Hinv, R, K, K, sym, qmin, qmax, group_size, static_groups, col_group, scales,
                                      zeros, Wout, losses, blocksize);
}

int gptq_weight_transform_cols(float* W, const float* Hinv, int64_t R, int64_t K, int64_t n_quant, int sym, float qmin,
                               float qmax, int64_t group_size, int static_groups, const int32_t* col_group,
                               float* scales, float* zeros, float* Wout, float* losses, int blocksize) {
    const int per_channel = group_size <= 0;
    const int static_mode = static_groups || per_channel;
    const int64_t ng = per_channel ? 1 : (K + group_size - 1) / group_size;
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t r = 0; r < R; ++r) {
        float* w = W + r * K;
        float* err = (float*)malloc(sizeof(float) * blocksize);
        float* w1 = (float*)malloc(sizeof(float) * blocksize);
        float s = 1.0f, z = 0.0f;
        for (int64_t i1 = 0; i1 < n_quant; i1 += blocksize) {
            const int64_t i2 = i1 + blocksize < n_quant ? i1 + blocksize : n_quant;
            const int count = (int)(i2 - i1);
            memcpy(w1, w + i1, sizeof(float) * count);      /* W1 = W[:, i1:i2].clone() */
            for (int i = 0; i < count; ++i) {
                const int64_t col = i1 + i;
                const float d = Hinv[col * K + col];
                if (!static_mode) {
                    if (col % group_size == 0) {
                        /* search_column_qparams on W[:, col : min(col+g, K)]: NOTE the reference reads W,
                           not W1 (gptq.py:216); inside the current block W is stale w.r.t. the in-block
                           updates only when a group starts mid-block (group_size < blocksize). */
                        int64_t e = col + group_size < n_quant ? col + group_size : n_quant;
                        float mn = INFINITY, mx = -INFINITY;
                        for (int64_t c = col; c < e; ++c) {
                            float v = w[c];
                            mn = fminf(mn, v);
                            mx = fmaxf(mx, v);
                        }
                        qp_t q = qparams_f32(mn, mx, sym, qmin, qmax);
                        s = q.s;
                        z = q.z;
                        scales[r * ng + col / group_size] = s;
                        if (zeros) zeros[r * ng + col / group_size] = z;
                    }
                } else {
                    const int64_t g = per_channel ? 0 : col_group[col];
                    s = scales[r * ng + g];
                    z = zeros ? zeros[r * ng + g] : 0.0f;
                }
                const float wv = w1[i];
                const float q = qdq_f32(wv, s, z, qmin, qmax);
                Wout[r * K + col] = wv;                                          /* tmp1[:, i] = w */
                const float diff = wv - q;
                if (losses) losses[r * K + col] = (diff * diff) / (2.0f * (d * d));
                const float e1 = diff / d;
                err[i] = e1;
                for (int j = i; j < count; ++j) {                               /* W1[:, i:] -= e1 * Hinv1[i, i:] */
                    const float t = e1 * Hinv[col * K + i1 + j];
                    w1[j] = w1[j] - t;
                }
            }
            for (int64_t j = i2; j < K; ++j) {                                   /* W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:] */
                float acc = 0.0f;
                for (int k = 0; k < count; ++k) acc = fmaf(err[k], Hinv[(i1 + k) * K + j], acc);
                w[j] = w[j] - acc;
            }
        }
        free(err);
        free(w1);
    }
    return rc;
}

/* out[r][n] = chain_k fmaf(a[r][k], b[k][n], acc) from +0: the sgemm model used to pin MKL's order */
void mm_chain(const float* a, const float* b, float* out, int64_t R, int64_t Kc, int64_t N) {
#pragma omp parallel for
    for (int64_t r = 0; r < R; ++r)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int64_t k = 0; k < Kc; ++k) acc = fmaf(a[r * Kc + k], b[k * N + n], acc);
            out[r * N + n] = acc;
        }
}
