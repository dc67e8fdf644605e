// fp8_math.h — OCP e4m3fn conversions shared by fp8_pack.hip and fp8_block.hip. Rounding is pinned to torch.float8_e4m3fn's
// cast (round to nearest even, no saturation: a value that rounds above 448 becomes NaN).
#pragma once
#include "common.h"

namespace llmc {

// fp32 -> e4m3fn bits with torch's semantics
__device__ __forceinline__ uint8_t f32_to_e4m3fn(float x) {
    const uint32_t b = __float_as_uint(x);
    const uint32_t sign = (b >> 24) & 0x80u;
    const uint32_t ab = b & 0x7fffffffu;
    if (ab > 0x7f800000u) return (uint8_t)(sign | 0x7f);  // NaN
    const float ax = __uint_as_float(ab);
    if (ax < 0.015625f) {  // below 2^-6: subnormal grid 2^-9 (rint = RNE); 8 -> smallest normal
        const uint32_t m = (uint32_t)rintf(ax * 512.0f);
        return (uint8_t)(sign | m);
    }
    // round the fp32 mantissa to 3 bits, RNE, carry propagates into the exponent
    uint32_t r = ab + 0x7ffffu + ((ab >> 20) & 1u);
    r &= 0xfff00000u;
    if (r > 0x43e00000u) return (uint8_t)(sign | 0x7f);  // > 448 after rounding (incl. inf) -> NaN
    const uint32_t e = (r >> 23) - 127 + 7;
    const uint32_t m = (r >> 20) & 7u;
    return (uint8_t)(sign | (e << 3) | m);
}

// Two values at once on v_cvt_pk_fp8_f32 (gfx950: OCP e4m3fn, round to nearest even). Measured against the routine above over
// 4 M bit patterns (tools/probes/fp8_mfma_probe.hip, profiles/r03_fp8_mfma_probe.txt): identical for every finite |x| <= 464
// (= everything that rounds to a finite e4m3 value, subnormals included); NaN and overflow differ (sign of the NaN code), so
// those take the routine above. Result: code of x in bits 0-7, of y in bits 8-15.
__device__ __forceinline__ uint32_t f32x2_to_e4m3fn(float x, float y) {
    if (__builtin_expect(!(fabsf(x) <= 464.0f) || !(fabsf(y) <= 464.0f), 0))
        return (uint32_t)f32_to_e4m3fn(x) | ((uint32_t)f32_to_e4m3fn(y) << 8);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x, y, 0, false) & 0xffffu;
}

__device__ __forceinline__ float e4m3fn_to_f32(uint8_t v) {
    const uint32_t e = (v >> 3) & 0xf, m = v & 7;
    float r;
    if ((v & 0x7f) == 0x7f) r = __uint_as_float(0x7fc00000u);
    else if (e == 0) r = (float)m * 0.001953125f;
    else r = __uint_as_float(((e - 7 + 127) << 23) | (m << 20));
    // sign by bit: a negative zero must stay negative ((q - 0) * s = -0.0 in the reference)
    return __uint_as_float(__float_as_uint(r) | ((uint32_t)(v & 0x80) << 24));
}

// ---- e5m2 (torch.float8_e5m2: IEEE-style binary8, round to nearest even, overflow -> inf) ------------------------------
__device__ __forceinline__ uint8_t f32_to_e5m2(float x) {
    const uint32_t b = __float_as_uint(x);
    const uint32_t sign = (b >> 24) & 0x80u;
    const uint32_t ab = b & 0x7fffffffu;
    if (ab > 0x7f800000u) return (uint8_t)(sign | 0x7f);  // NaN
    const float ax = __uint_as_float(ab);
    if (ax < 6.103515625e-05f) {   // below 2^-14: subnormal grid 2^-16 (rint = RNE); 4 -> smallest normal
        const uint32_t m = (uint32_t)rintf(ax * 65536.0f);
        return (uint8_t)(sign | m);
    }
    uint32_t r = ab + 0xfffffu + ((ab >> 21) & 1u);
    r &= 0xffe00000u;
    if (r > 0x47600000u) return (uint8_t)(sign | 0x7c);   // > 57344 after rounding (incl. inf) -> inf
    const uint32_t e = (r >> 23) - 127 + 15;
    const uint32_t m = (r >> 21) & 3u;
    return (uint8_t)(sign | (e << 2) | m);
}
__device__ __forceinline__ float e5m2_to_f32(uint8_t v) {
    const uint32_t e = (v >> 2) & 0x1f, m = v & 3;
    float r;
    if (e == 0x1f) r = m ? __uint_as_float(0x7fc00000u) : __uint_as_float(0x7f800000u);
    else if (e == 0) r = (float)m * 1.52587890625e-05f;
    else r = __uint_as_float(((e - 15 + 127) << 23) | (m << 21));
    return __uint_as_float(__float_as_uint(r) | ((uint32_t)(v & 0x80) << 24));
}

// ---- qtorch.quant.float_quantize(x, E, M, rounding='nearest') (subnormals=True, saturate=True), restated from QPyTorch
// 0.3.0 (quant_cpu.cpp:float_quantize_nearest, bit_helper.cpp) — the arithmetic llmc's FloatQuantizer.quant runs
// (quant.py:1061-1072); oracle/quant_ref.py:qtorch_float_quantize is the same restatement in numpy, with the list of
// differences from torch's own cast. Returns the quantized VALUE (every result is exactly representable in the OCP /
// IEEE 8-bit type of the same widths, so the real-quant cast that follows in the reference is exact).
template <int E, int M> __device__ __forceinline__ float qtorch_quantize(float x) {
    const uint32_t b = __float_as_uint(x);
    const uint32_t sign = b & 0x80000000u;
    const int t_exp = (int)((b & 0x7fffffffu) >> 23) - 127;
    constexpr int min_exp = -((1 << (E - 1)) - 2);
    constexpr uint32_t mask = (1u << (23 - M)) - 1u, half = 1u << (22 - M);
    if (t_exp < min_exp) {                       // the target format's subnormal range: round at 2^min_exp's spacing
        const float shift = __uint_as_float(((uint32_t)(127 + min_exp) << 23) | sign);
        const float val = opaque_f32(x + shift);
        const uint32_t q = (__float_as_uint(val) + half) & ~mask;
        return __uint_as_float(q) - shift;
    }
    uint32_t q = (b + half) & ~mask;             // nearest, ties away from zero
    constexpr uint32_t max_store = (1u << (E - 1)) - 1u + 127u;      // the top exponent code is kept for infinity
    if (q != 0 && ((q & 0x7fffffffu) >> 23) > max_store) q = sign | (max_store << 23) | (((1u << M) - 1u) << (23 - M));
    return __uint_as_float(q);
}

// qtorch_quantize<4, 3> without control flow for a finite, non-NaN x (both ranges evaluated, one selected): the normal range
// as above with the saturation as a clamp to +-240 (256 is the next representable magnitude: nothing lies between); below 2^-6
// the multiples of 2^-9 with ties away, formed exactly like the routine above (|x| + 2^-6 in fp32, rounded on its bits, shifted
// back; a zero result is +0). A NaN takes the routine
// above. tests/test_fp8_fast_gpu.py and the fp8 goldens run both forms on the same data.
__device__ __forceinline__ float qtorch_e4m3_select(float x) {
    const uint32_t u = __float_as_uint(x), a = u & 0x7fffffffu;
    if (__builtin_expect(a > 0x7f800000u, 0)) return qtorch_quantize<4, 3>(x);
    const float n = __builtin_amdgcn_fmed3f(__uint_as_float((u + 0x80000u) & 0xfff00000u), -240.0f, 240.0f);
    // QPyTorch's own two roundings: fl32(|x| + 2^-6) first (spacing 2^-29 there), then ties-away at 2^-9's spacing on
    // the sum's bits — an input within 2^-30 below a (k + 0.5) 2^-9 midpoint is carried over it by the first one
    const float va = opaque_f32(__uint_as_float(a) + 0.015625f);
    const float r = __uint_as_float((__float_as_uint(va) + 0x80000u) & 0xfff00000u) - 0.015625f;
    const float d = __uint_as_float(__float_as_uint(r) | (u & 0x80000000u)) + 0.0f;
    return a < 0x3c800000u ? d : n;
}

// One element of a FloatQuantizer cast. fmt: 0 = e4m3, 1 = e5m2. sem: 0 = the dtype cast of torch (float8_e4m3fn /
// float8_e5m2: round to nearest even), 1 = qtorch.float_quantize. Returns the code; *val receives the decoded value.
__device__ __forceinline__ uint8_t fp8_encode(float t, int fmt, int sem, float* val) {
    if (sem) {
        const float v = fmt ? qtorch_quantize<5, 2>(t) : qtorch_quantize<4, 3>(t);
        *val = v;
        return fmt ? f32_to_e5m2(v) : f32_to_e4m3fn(v);
    }
    const uint8_t q = fmt ? f32_to_e5m2(t) : f32_to_e4m3fn(t);
    *val = fmt ? e5m2_to_f32(q) : e4m3fn_to_f32(q);
    return q;
}
__device__ __forceinline__ float fp8_format_max(int fmt) { return fmt ? 57344.0f : 448.0f; }

}  // namespace llmc
