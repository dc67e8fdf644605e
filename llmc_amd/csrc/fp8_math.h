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

}  // namespace llmc
