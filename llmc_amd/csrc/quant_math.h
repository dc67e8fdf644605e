// quant_math.h — the scalar arithmetic of llmc's IntegerQuantizer, stated once for every kernel.
// Mirrors llmc/compression/quantization/quant.py:545-559 (get_qparams) and :699-717 (quant/dequant);
// oracle/quant_ref.py restates the same chain in numpy and is pinned to the reference by tests/golden.
#pragma once
#include "common.h"

namespace llmc {

struct QParams {
    float s;
    float z;
};

// min/max are values of the tensor dtype `dt` (exact in fp32). Every op rounds to `dt` like ATen does.
__device__ __forceinline__ QParams qparams_from_minmax(float mn, float mx, int dt, int sym, int round_zp,
                                                       float qmin, float qmax) {
    QParams q;
    const float eps = rnd(1e-5f, dt);  // clamp(min=1e-5): the scalar is cast to the tensor dtype
    if (sym) {
        float a = fmaxf(fabsf(mx), fabsf(mn));  // quant.py:550
        a = fmaxf(a, eps);                      // :551
        q.s = rnd(a / rnd(qmax, dt), dt);       // :552
        q.z = 0.0f;                             // :553
    } else {
        float d = rnd(mx - mn, dt);  // :555
        d = fmaxf(d, eps);
        q.s = rnd(d / rnd(qmax - qmin, dt), dt);
        float r = rnd(mn / q.s, dt);
        if (round_zp) {
            r = rintf(r);                  // torch.round: half-to-even, exact in dt
            float z = rnd(qmin - r, dt);   // :556
            q.z = fminf(fmaxf(z, qmin), qmax);
        } else {
            q.z = rnd(qmin - r, dt);  // :557-558
        }
    }
    return q;
}

// quant (quant.py:699-707, round_zp=True branch). p1 = promote(wdt, sdt); p2 = promote(p1, zdt).
__device__ __forceinline__ float quant_code(float x, float s, float z, int p1, int p2, float qmin,
                                            float qmax) {
    float t = rnd(x / s, p1);
    t = rintf(t);
    t = rnd(t + z, p2);
    return fminf(fmaxf(t, qmin), qmax);
}
// dequant (quant.py:709-712)
__device__ __forceinline__ float dequant_code(float q, float s, float z, int p2) {
    float t = rnd(q - z, p2);
    return rnd(t * s, p2);
}

}  // namespace llmc
