/*
 * spqr_canon.c — C restatement of SpQR.weight_transform (llmc/compression/quantization/spqr.py:185-254) with its
 * helpers `outliers` (spqr.py:186-203), get_group_qparams (spqr.py:323-345) and the round_zp=False quantizer
 * (quant.py:545-559, 699-717), in the reference's fp32 operation order. TEST INFRASTRUCTURE (oracle): the checker of
 * tests/, never the product path.
 *
 * As executed by the reference (asymmetric weights; a symmetric weight quantizer crashes in get_group_qparams, its
 * zero point being 0-dim):
 *   per group start (i % g == 0), G = W[:, i:i+g] (current running weights):
 *     detection (unless simplified_outliers or threshold == inf), Q = per-row asym, round_zp=False:
 *       Base   = sum_k ((qdq_all(G_k) - G_k) / d_k)^2            d_k = Hinv[k][k]
 *       Loo_j  = sum_{k != j} ((qdq_without_j(G_k) - G_k) / d_k)^2
 *       M_j    = (Base - Loo_j) > threshold ; mean = sum G(1-M) / max(sum(1-M), 1) ; G' = G(1-M) + mean M
 *     (s, z) = asym round_zp=False qparams of G' (or G);
 *     second level: the scale / zero quantizers see s, z as [R, 1] tensors, so reshape_tensor leaves them alone
 *     (last dim 1 < group_size) and min == max per row: ss = 1e-5/(qmax-qmin), zs = qmin - s/ss, code
 *     round(s/ss + zs) = 0, dequant (0 - zs) * ss = fl(fl(s/ss) * ss): the stored scale is s after one division and
 *     one multiplication by 1e-5/7, the zero point likewise.
 *   per column: q = (clamp(round(w / max(s,1e-9) + z)) - z) * s ; err = (w - q) / d ;
 *     mask = err^2 > threshold ; masked columns keep w (err = (w - w) / d) ; tmp = w ; loss = err^2 ;
 *     W[:, i+1:i2] -= fl(err * Hinv[i][i+1:i2])
 *   per block: W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]   (fma chain from +0, k ascending, as in gptq_canon.c)
 * The sums over a group run in ascending k here; torch's CPU reduction order over a contiguous last dim differs in
 * the last bits, so the goldens pin the detection by tolerance (tests/test_oracle_golden.py), everything after the
 * mask bit-exactly given the same mask.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline void qp_asym_nr(float mn, float mx, float qmin, float qmax, float* s, float* z) {
    float d = mx - mn;
    if (d < 1e-5f) d = 1e-5f;
    *s = d / (qmax - qmin);
    *z = qmin - (mn / *s);
}

static inline float qdq_nr(float x, float s, float z, float qmin, float qmax) {
    const float sc = s < 1e-9f ? 1e-9f : s;
    float t = x / sc;
    t = t + z;
    t = rintf(t);
    t = fminf(fmaxf(t, qmin), qmax);
    t = t - z;
    return t * s;
}

/* a [R,1] tensor through a per_group / per_channel round_zp=False quantizer: fl(fl(v / ss) * ss) */
static inline float second_level(float v, float lqmin, float lqmax) {
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

/*
 * W [R,K] in/out (running weights); Hinv [K,K] upper factor; group size g divides blocksize; threshold (already
 * relative_threshold * outlier_scale, as fp32) may be INFINITY; detect = !simplified_outliers.
 * Outputs: Wout (tmp) [R,K], losses [R,K], mask [R,K] uint8, scales / zeros [R, K/g] (second-level values).
 * sq*: qmin/qmax of the scale quantizer, zq*: of the zero quantizer.
 */
int spqr_weight_transform(float* W, const float* Hinv, int64_t R, int64_t K, float qmin, float qmax, int64_t g,
                          float threshold, int detect, float sqmin, float sqmax, float zqmin, float zqmax,
                          float* scales, float* zeros, float* Wout, float* losses, uint8_t* mask, int blocksize) {
    if (g <= 0 || blocksize % g != 0 || K % g != 0) return -1;
    const int64_t ng = K / g;
    const int use_detect = detect && !isinf(threshold);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t r = 0; r < R; ++r) {
        float* w = W + r * K;
        float* err = (float*)malloc(sizeof(float) * blocksize);
        float* gm = (float*)malloc(sizeof(float) * g);
        float s = 1.0f, z = 0.0f;
        for (int64_t i1 = 0; i1 < K; i1 += blocksize) {
            const int64_t i2 = i1 + blocksize < K ? i1 + blocksize : K;
            const int count = (int)(i2 - i1);
            for (int i = 0; i < count; ++i) {
                const int64_t col = i1 + i;
                const float d = Hinv[col * K + col];
                if (col % g == 0) {
                    const float* G = w + col;
                    float mn = INFINITY, mx = -INFINITY;
                    if (use_detect) {
                        for (int k = 0; k < g; ++k) { mn = fminf(mn, G[k]); mx = fmaxf(mx, G[k]); }
                        float bs, bz;
                        qp_asym_nr(mn, mx, qmin, qmax, &bs, &bz);
                        float base = 0.0f;
                        for (int k = 0; k < g; ++k) {
                            const float e = (qdq_nr(G[k], bs, bz, qmin, qmax) - G[k]) / Hinv[(col + k) * K + col + k];
                            base = base + e * e;
                        }
                        float sum_keep = 0.0f, n_keep = 0.0f;
                        for (int j = 0; j < g; ++j) {
                            float lmn = INFINITY, lmx = -INFINITY;
                            for (int k = 0; k < g; ++k)
                                if (k != j) { lmn = fminf(lmn, G[k]); lmx = fmaxf(lmx, G[k]); }
                            float ls, lz;
                            qp_asym_nr(lmn, lmx, qmin, qmax, &ls, &lz);
                            float loo = 0.0f;
                            for (int k = 0; k < g; ++k)
                                if (k != j) {
                                    const float e = (qdq_nr(G[k], ls, lz, qmin, qmax) - G[k]) / Hinv[(col + k) * K + col + k];
                                    loo = loo + e * e;
                                }
                            const float m = (base - loo) > threshold ? 1.0f : 0.0f;
                            gm[j] = m;
                            sum_keep = sum_keep + G[j] * (1.0f - m);
                            n_keep = n_keep + (1.0f - m);
                        }
                        const float mean = sum_keep / (n_keep < 1.0f ? 1.0f : n_keep);
                        mn = INFINITY; mx = -INFINITY;
                        for (int k = 0; k < g; ++k) {
                            const float v = G[k] * (1.0f - gm[k]) + mean * gm[k];
                            mn = fminf(mn, v);
                            mx = fmaxf(mx, v);
                        }
                    } else {
                        for (int k = 0; k < g; ++k) { mn = fminf(mn, G[k]); mx = fmaxf(mx, G[k]); }
                    }
                    float s1, z1;
                    qp_asym_nr(mn, mx, qmin, qmax, &s1, &z1);
                    s = second_level(s1, sqmin, sqmax);
                    z = second_level(z1, zqmin, zqmax);
                    scales[r * ng + col / g] = s;
                    zeros[r * ng + col / g] = z;
                }
                const float wv = w[col];
                const float q = qdq_nr(wv, s, z, qmin, qmax);
                float e1 = (wv - q) / d;
                uint8_t mk = 0;
                if (!isinf(threshold)) {
                    mk = (e1 * e1) > threshold;
                    const float M = mk ? 1.0f : 0.0f;
                    const float newq = q * (1.0f - M) + wv * M;
                    e1 = (wv - newq) / d;
                }
                mask[r * K + col] = mk;
                Wout[r * K + col] = wv;
                losses[r * K + col] = e1 * e1;
                err[i] = e1;
                for (int j = i + 1; j < count; ++j) {
                    const float t = e1 * Hinv[col * K + i1 + j];
                    w[i1 + j] = w[i1 + j] - t;
                }
            }
            for (int64_t j = i2; j < K; ++j) {
                float acc = 0.0f;
                for (int k = 0; k < count; ++k) acc = fmaf(err[k], Hinv[(i1 + k) * K + j], acc);
                w[j] = w[j] - acc;
            }
        }
        free(err);
        free(gm);
    }
    return 0;
}
