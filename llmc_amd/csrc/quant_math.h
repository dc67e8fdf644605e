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


// ---- division by a value that is fixed for many elements (a group's scale) --------------------------------------
// `x / s` compiles to hipcc's IEEE sequence: v_div_scale x2, rcp, 2 fma (reciprocal refinement), mul, 3 fma, v_div_fmas,
// v_div_fixup. v_div_scale is the identity when the operands are "plain" and v_div_fixup only acts on zero / inf /
// nan / denormal operands, so with the refined reciprocal hoisted the quotient is the 5-op tail below, bit for bit.
// Plain: divisor in [2^-40, 2^40) and |x| < 2^40. (|x| below 2^-103 is scaled by the full sequence; the tail then
// differs in the last bits of a quotient below 2^-63, which every caller rounds to an integer: still 0.)
__device__ __forceinline__ bool plain_pos(float x) { return (__float_as_uint(x) - 0x2B800000u) < 0x28000000u; }
__device__ __forceinline__ float rcp_refined(float d) {
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e0 = fmaf(-d, y0, 1.0f);
    return fmaf(e0, y0, y0);
}
__device__ __forceinline__ float div_tail(float n, float d, float y) {
    const float q0 = n * y;
    const float e1 = fmaf(-d, q0, n);
    const float q1 = fmaf(e1, y, q0);
    const float e2 = fmaf(-d, q1, n);
    return fmaf(e2, y, q1);
}
struct Divisor {
    float s, y;
    bool fast;
};
// absmax: an upper bound of |x| over the elements that will be divided (the group's max(|min|, |max|))
__device__ __forceinline__ Divisor make_divisor(float s, float absmax) {
    Divisor d;
    d.s = s;
    d.y = rcp_refined(s);
    d.fast = plain_pos(s) && absmax < 1.099511627776e12f;   // 2^40; false for inf / nan bounds
    return d;
}
__device__ __forceinline__ float div_by(float x, const Divisor& d) { return d.fast ? div_tail(x, d.s, d.y) : x / d.s; }

// quant (quant.py:699-707, round_zp=True branch). p1 = promote(wdt, sdt); p2 = promote(p1, zdt).
__device__ __forceinline__ float quant_code(float x, float s, float z, int p1, int p2, float qmin,
                                            float qmax) {
    float t = rnd(x / s, p1);
    t = rintf(t);
    t = rnd(t + z, p2);
    return fminf(fmaxf(t, qmin), qmax);
}
// same with the group's hoisted divisor
__device__ __forceinline__ float quant_code(float x, const Divisor& d, float z, int p1, int p2, float qmin,
                                            float qmax) {
    float t = rnd(div_by(x, d), p1);
    t = rintf(t);
    t = rnd(t + z, p2);
    return fminf(fmaxf(t, qmin), qmax);
}
// quant, round_zp=False branch (quant.py:702-707): round(x / s.clamp_min(1e-9) + z); d is the divisor of the CLAMPED scale
__device__ __forceinline__ float quant_code_fz(float x, const Divisor& d, float z, int p1, int p2, float qmin,
                                               float qmax) {
    float t = rnd(div_by(x, d), p1);
    t = rnd(t + z, p2);
    t = rintf(t);
    return fminf(fmaxf(t, qmin), qmax);
}
// dequant (quant.py:709-712)
__device__ __forceinline__ float dequant_code(float q, float s, float z, int p2) {
    float t = rnd(q - z, p2);
    return rnd(t * s, p2);
}

}  // namespace llmc
