// fp8_pack.hip — K11 FP8 (OCP e4m3fn) weight quantization and the AutoAWQ GEMM packer (K7b).
//   llmc_fp8_quant      FloatQuantizer sym e4m3 / e5m2: scale = absmax.clamp(1e-5) / finfo.max, q = float_quantize(x / scale)
//                       (quant.py:545-553, 983-1003, 1061-1072, 1195-1221) with qtorch's arithmetic restated (fp8_math.h:
//                       qtorch_quantize; the library is a third-party dependency absent from the reference tree), or
//                       torch's own dtype cast (float8_e4m3fn / float8_e5m2, RNE) for the callers that end in one.
//   llmc_pack_awq_gemm  module_utils.py:1004-1065.
#include <stdlib.h>
#include "common.h"
#include "fp8_math.h"

namespace llmc {

static constexpr int FB = 256;
// k_fp8_cast: vectors per thread and turn (U), their layout and the grid cap. Measured on a 14336 x 4096 weight
// (tools/probes/build_fp8_variants.sh, profiles/r04_fp8_cast_ab.txt): ONE vector per thread is the fastest — the kernel is
// bound by its VALU work, not by loads in flight (U = 2 equal, U = 4 / 8 and a smaller grid slower). Lab builds override.
#ifndef FP8_U
#define FP8_U 1
#endif
#ifndef FP8_PATTERN
#define FP8_PATTERN 1
#endif
#ifndef FP8_GRID_CAP
#define FP8_GRID_CAP 8192
#endif

// amax[row] = clamp(absmax, 1e-5) in dt (from llmc_minmax_qparams with qmax = 1). scale = amax / 448 in the
// scales' dtype sdt: ATen promotes the 0-dim per-tensor absmax (dt) / 0-dim fp32 qmax to fp32, but keeps dt for
// the per-channel [R,1] absmax.
// mode: bit 0 fake (write dequantized values), bits 4-5 format (0 e4m3, 1 e5m2), bit 8 semantics (0 torch's dtype cast,
// 1 qtorch.float_quantize: fp8_math.h)
static constexpr int FP8_FAKE = 1, FP8_FMT_SHIFT = 4, FP8_QTORCH = 0x100;
static constexpr int FP8_NO_PACKED16 = 0x400;    // A/B: the float form of the division-free path (round 4) instead of the packed 16-bit one
template <typename T>
__device__ __forceinline__ void fp8_one(float w, float s, int tdt, int mode, int DT, T* of, uint8_t* ob) {
    const float t = rnd(rnd(w / s, tdt) + 0.0f, tdt);          // tensor / scales + zeros
    float v;
    const uint8_t q = fp8_encode(t, (mode >> FP8_FMT_SHIFT) & 3, mode & FP8_QTORCH, &v);
    if (mode & FP8_FAKE) *of = from_f32<T>(opaque_f32(v * s));   // fp32 product, one rounding to dt
    else *ob = q;
}
// two elements: one hardware conversion (fp8_math.h) on the e4m3 cast path
template <typename T>
__device__ __forceinline__ void fp8_two(float w0, float w1, float s, int tdt, int mode, T* of, uint8_t* ob) {
    const int fake = mode & FP8_FAKE;
    if ((mode & ~FP8_FAKE) == FP8_QTORCH) {      // e4m3 with qtorch's rounding: the quantized values are exact e4m3fn numbers,
        const float v0 = qtorch_quantize<4, 3>(rnd(rnd(w0 / s, tdt) + 0.0f, tdt));     // so the hardware conversion is exact
        const float v1 = qtorch_quantize<4, 3>(rnd(rnd(w1 / s, tdt) + 0.0f, tdt));
        if (fake) {
            of[0] = from_f32<T>(opaque_f32(v0 * s));
            of[1] = from_f32<T>(opaque_f32(v1 * s));
        } else {
            const uint32_t c = f32x2_to_e4m3fn(v0, v1);
            ob[0] = (uint8_t)c;
            ob[1] = (uint8_t)(c >> 8);
        }
        return;
    }
    if (mode & ~FP8_FAKE) {
        fp8_one<T>(w0, s, tdt, mode, 0, &of[0], &ob[0]);
        fp8_one<T>(w1, s, tdt, mode, 0, &of[1], &ob[1]);
        return;
    }
    const float t0 = rnd(rnd(w0 / s, tdt) + 0.0f, tdt), t1 = rnd(rnd(w1 / s, tdt) + 0.0f, tdt);
    const uint32_t c = f32x2_to_e4m3fn(t0, t1);
    const uint8_t q0 = (uint8_t)c, q1 = (uint8_t)(c >> 8);
    if (fake) {
        of[0] = from_f32<T>(opaque_f32(e4m3fn_to_f32(q0) * s));
        of[1] = from_f32<T>(opaque_f32(e4m3fn_to_f32(q1) * s));
    } else {
        ob[0] = q0;
        ob[1] = q1;
    }
}

// ---- eight 16-bit elements at once, without the division ---------------------------------------------------------------
// t = rnd_dt(fl32(w / s)) is what the reference's `tensor / scales` leaves (ATen divides in fp32, then rounds to the tensor
// dtype). q = w * fl32(1 / s) is within 3 fp32 ulps of fl32(w / s), so both round to the same dt value unless q lies within
// 4 ulps of a dt rounding boundary (9 of 65536 bf16 patterns, 9 of 8192 fp16 patterns) — those lanes, f16 results outside
// the normal range, non-finite values and, for the dtype cast, |t| > 464 take the division and the general encoder instead (FP8_EXACT_DIV in `mode` sends everything there: the A/B switch
// of tests/test_fp8_fast_gpu.py). e4m3 only; qtorch semantics on the fast path = (bits + 0x80000) & ~0xfffff, saturated to
// 240 from 256 upwards (fp8_math.h:qtorch_quantize), then the exact hardware conversion of the on-grid values.
static constexpr int FP8_EXACT_DIV = 0x200;

// Round 6: eight bf16 elements -> eight e4m3 CODES with qtorch's rounding on 16-bit lanes (two elements per packed
// instruction), about 15 VALU instructions per element instead of 21 (the cast is VALU-bound: 42 us against 21 us of HBM time
// for a 14336 x 4096 weight). t = bf16(w * fl(1 / s)) as above (same tie guard, on the quotients' low halves packed into one
// dword). On t's bf16 pattern m (sign cleared):
//   |t| >= 2^-6: ties-away on the 3-bit mantissa is (m + 8) >> 4 = 8 * exponent + mantissa, the e4m3 code is that minus
//                8 * 120, saturated at 0x77 (240: qtorch keeps the top exponent code for infinity, 248 and above land there);
//   |t| <  2^-6: the format's subnormal range — spacing 2^-9, and the code of k * 2^-9 IS k: floor(|t| * 512 + 0.5), exact for
//                an 8-bit significand (the same formula as the float form below);
// a per-half mask picks one, the sign goes to bit 7 unless the code is 0 (qtorch's x - x = +0). Inf / NaN patterns of t and
// quotients within 4 ulps of a rounding boundary leave through the general encoder like before. Bit-identical to the float
// form and to the division form (tests/test_fp8_fast_gpu.py, fp8 goldens).
__device__ __forceinline__ bool fp8_codes8_bf16_qtorch(const uint4 raw, float rs, uint2* ob) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const uint32_t word[4] = {raw.x, raw.y, raw.z, raw.w};
    u16x2 tie = {0xffff, 0xffff}, big = {0, 0};
    uint32_t cw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f2 w = {__uint_as_float(word[p] << 16), __uint_as_float(word[p] & 0xffff0000u)};
        const f2 q = __builtin_elementwise_fma(w, (f2){rs, rs}, (f2){0.0f, 0.0f});      // w * (1 / s) + 0: -0 becomes +0
        const uint32_t lows = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), 0x05040100u);
        tie = __builtin_elementwise_min(tie, (u16x2)(__builtin_bit_cast(u16x2, lows) - (u16x2){0x7ffc, 0x7ffc}));
        const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf2));     // v_cvt_pk_bf16_f32: RNE
        const uint32_t m = pk & 0x7fff7fffu;
        const u16x2 mv = __builtin_bit_cast(u16x2, m);
        big = __builtin_elementwise_max(big, mv);
        u16x2 r = (u16x2)((u16x2)(mv + (u16x2){8, 8}) >> (u16x2){4, 4});
        r = (u16x2)(__builtin_elementwise_min(r, (u16x2){1079, 1079}) - (u16x2){960, 960});
        const uint32_t k0 = (uint32_t)__builtin_fmaf(__builtin_fabsf(__uint_as_float(pk << 16)), 512.0f, 0.5f);
        const uint32_t k1 = (uint32_t)__builtin_fmaf(__builtin_fabsf(__uint_as_float(pk & 0xffff0000u)), 512.0f, 0.5f);
        const uint32_t kk = __builtin_amdgcn_perm(k1, k0, 0x05040100u);      // the low halves (a large |t| * 512 does not fit 16 bits: its half is not selected)
        const uint32_t sub = __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, (u16x2)(mv - (u16x2){0x3c80, 0x3c80})) >> (i16x2){15, 15}));
        uint32_t c = (sub & kk) | (~sub & __builtin_bit_cast(uint32_t, r));
        const uint32_t nz = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, c) + (u16x2){0x7f, 0x7f}));   // bit 7: code != 0
        c |= (pk >> 8) & nz & 0x00800080u;
        cw[p] = c;
    }
    if (tie[0] <= 8 || tie[1] <= 8 || big[0] >= 0x7f80 || big[1] >= 0x7f80) return false;
    ob->x = __builtin_amdgcn_perm(cw[1], cw[0], 0x06040200u);
    ob->y = __builtin_amdgcn_perm(cw[3], cw[2], 0x06040200u);
    return true;
}

template <typename T>
__device__ __forceinline__ bool fp8_fast8(const uint4 raw, float s, float rs, int mode, T (&of)[8], uint2* ob) {
    constexpr int DT = dt_of<T>::value;
    const bool qt = mode & FP8_QTORCH;
    if constexpr (DT == LLMC_BF16) {
        if (qt && !(mode & FP8_FAKE) && !(mode & FP8_NO_PACKED16)) return fp8_codes8_bf16_qtorch(raw, rs, ob);
    }
    const uint32_t word[4] = {raw.x, raw.y, raw.z, raw.w};
    // guards as running minima / maxima (one v_min3 / v_max3 per pair instead of compare + or per element):
    //   tie   = min of ((q bits & low mask) - (midpoint - 4)) as unsigned: <= 8 means within 4 ulps of a rounding boundary
    //   tiny16 = min of (|q| bits - 1) as unsigned: a nonzero fp16 quotient below the fp16 normal range
    //   big   = max of |t| bits: above 464 [cast], at the top of the fp16 range
    uint32_t special = 0, tie = 0xffffffffu, tiny16 = 0xffffffffu, big = 0;
    float v[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // an all-ones exponent in either half (inf / nan): the general path keeps their special cases
        constexpr uint32_t EXPM = DT == LLMC_BF16 ? 0x7f807f80u : 0x7c007c00u, EXPL = DT == LLMC_BF16 ? 0x00800080u : 0x04000400u;
        special |= ((word[p] & EXPM) + EXPL) & 0x80008000u;
        float q0, q1;                                                     // w * (1 / s) + 0: the `+ zeros` of the reference turns -0 into +0
        if constexpr (DT == LLMC_BF16) {
            q0 = __builtin_fmaf(__uint_as_float(word[p] << 16), rs, 0.0f);
            q1 = __builtin_fmaf(__uint_as_float(word[p] & 0xffff0000u), rs, 0.0f);
        } else {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 hv = __builtin_bit_cast(h2, word[p]);
            q0 = __builtin_fmaf((float)hv[0], rs, 0.0f);
            q1 = __builtin_fmaf((float)hv[1], rs, 0.0f);
        }
        const uint32_t b0 = __float_as_uint(q0), b1 = __float_as_uint(q1);
        float t0, t1;
        if constexpr (DT == LLMC_BF16) {
            tie = min(tie, min((b0 & 0xffffu) - 0x7ffcu, (b1 & 0xffffu) - 0x7ffcu));
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 qq = {q0, q1};
            const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(qq, bf2));     // v_cvt_pk_bf16_f32: RNE
            t0 = __uint_as_float(pk << 16);
            t1 = __uint_as_float(pk & 0xffff0000u);
        } else {
            tie = min(tie, min((b0 & 0x1fffu) - 0x0ffcu, (b1 & 0x1fffu) - 0x0ffcu));
            tiny16 = min(tiny16, min((b0 & 0x7fffffffu) - 1u, (b1 & 0x7fffffffu) - 1u));
            t0 = (float)(_Float16)q0;
            t1 = (float)(_Float16)q1;
        }
        const uint32_t u0 = __float_as_uint(t0), u1 = __float_as_uint(t1);
        const uint32_t a0 = u0 & 0x7fffffffu, a1 = u1 & 0x7fffffffu;
        big = max(big, max(a0, a1));
        if (qt) {
            // nearest with ties away on the 3-bit mantissa, then 256 and above -> 240 (nothing lies between)
            const float n0 = __builtin_amdgcn_fmed3f(__uint_as_float((u0 + 0x80000u) & 0xfff00000u), -240.0f, 240.0f);
            const float n1 = __builtin_amdgcn_fmed3f(__uint_as_float((u1 + 0x80000u) & 0xfff00000u), -240.0f, 240.0f);
            // below 2^-6 (the format's subnormal range) qtorch rounds x + sign * 2^-6 and subtracts the shift again: the
            // multiples of 2^-9, ties away, and a result of zero is +0 (x - x). Per-tensor scales put a good share of a weight
            // with outlier channels here, so this is evaluated for every element and selected, not branched to
            const float d0 = __uint_as_float(__float_as_uint(__builtin_floorf(__builtin_fmaf(__uint_as_float(a0), 512.0f, 0.5f)) * 0.001953125f) |
                                             (u0 & 0x80000000u)) + 0.0f;
            const float d1 = __uint_as_float(__float_as_uint(__builtin_floorf(__builtin_fmaf(__uint_as_float(a1), 512.0f, 0.5f)) * 0.001953125f) |
                                             (u1 & 0x80000000u)) + 0.0f;
            v[2 * p] = a0 < 0x3c800000u ? d0 : n0;
            v[2 * p + 1] = a1 < 0x3c800000u ? d1 : n1;
        } else {
            v[2 * p] = t0;
            v[2 * p + 1] = t1;
        }
    }
    bool slow = special != 0 || tie <= 8u;
    if constexpr (DT == LLMC_F16) slow = slow || tiny16 < 0x387fffffu || big >= 0x477fe000u;
    if (!qt) slow = slow || big > 0x43e80000u;                                // 464
    if (slow) return false;
    // four codes per dword straight from the converter (word select: low / high half of the destination)
    uint32_t lo = 0, hi = 0;
    lo = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], (int)lo, false);
    lo = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], (int)lo, true);
    hi = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], (int)hi, false);
    hi = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], (int)hi, true);
    if (mode & FP8_FAKE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint8_t c = (uint8_t)((k < 4 ? lo : hi) >> (8 * (k & 3)));
            of[k] = from_f32<T>(opaque_f32((qt ? v[k] : e4m3fn_to_f32(c)) * s));
        }
    } else {
        *ob = make_uint2(lo, hi);
    }
    return true;
}

template <typename T>
__global__ __launch_bounds__(FB) void k_fp8_cast(const T* __restrict__ W, const T* __restrict__ amax, int sdt,
                                                 void* __restrict__ scales, int static_scales, int64_t G, int64_t g,
                                                 int mode, void* __restrict__ out) {
    const int fake = mode & FP8_FAKE;
    const float fmax = fp8_format_max((mode >> FP8_FMT_SHIFT) & 3);
    constexpr int DT = dt_of<T>::value;
    constexpr int V = 16 / sizeof(T);
    const int pdt = promote(DT, sdt);
    const int tdt = (G == 1) ? DT : pdt;   // a 0-dim fp32 scale does not promote the [R,K] tensor, a [R,1] one does
    const bool vec = (g % V == 0) && (((uintptr_t)W & 15) == 0) && (((uintptr_t)out & 15) == 0);
    // the division-free form: 16-bit tensors whose quotient is rounded in their own dtype, e4m3, a scale whose reciprocal is an
    // ordinary number (checked per row below)
    const bool fast_kind = (V == 8) && tdt == DT && ((mode >> FP8_FMT_SHIFT) & 3) == 0 && !(mode & FP8_EXACT_DIV);
    const int64_t total = G * g;
    if (vec) {
        // U vectors per thread and turn, their loads issued together; a per_tensor call (G == 1) has no row arithmetic at all
        constexpr int U = FP8_U;
        // FP8_PATTERN 1: a workgroup walks U * 256 consecutive vectors per turn (stride between a thread's vectors = 256);
        // 0: a thread's vectors are a whole grid apart
        const int64_t nv = total / V;
        const int64_t stride = FP8_PATTERN ? (int64_t)FB : (int64_t)gridDim.x * FB;
        const int64_t first = FP8_PATTERN ? (int64_t)blockIdx.x * (U * FB) + threadIdx.x : (int64_t)blockIdx.x * FB + threadIdx.x;
        const int64_t turn = (int64_t)gridDim.x * FB * U;
        const int64_t vpr = g / V;                                       // vectors per row
        for (int64_t i0 = first; i0 < nv; i0 += turn) {
            uint4 raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * stride;
                if (i < nv) raw[u] = *reinterpret_cast<const uint4*>(W + i * V);
            }
            auto scale_of = [&](int64_t i) {
                const int64_t row = (G == 1) ? 0 : (nv < (int64_t)0xffffffffll ? (int64_t)((uint32_t)i / (uint32_t)vpr) : i / vpr);
                float s;
                if (static_scales) {
                    s = load_as_f32(scales, row, sdt);
                } else {
                    s = rnd(to_f32<T>(amax[row]) / fmax, sdt);
                    if (s == 0.0f) s = 1.0f;                          // scales[scales == 0] = 1 IN PLACE (quant.py:1062): the returned scale too
                    if (i * V == row * g) store_from_f32(scales, row, sdt, s);
                }
                return s == 0.0f ? 1.0f : s;                          // (static scales: the caller's tensor is left alone)
            };
            uint32_t todo = 0;                       // bit u: vector u is left to the general encoder below
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * stride;
                if (i < nv) {
                    bool done = false;
                    if constexpr (V == 8) {
                        if (fast_kind) {
                            const float s = scale_of(i);
                            if (s > 1e-30f && s < 1e30f) {
                                T of[V];
                                uint2 ob;
                                done = fp8_fast8<T>(raw[u], s, 1.0f / s, mode, of, &ob);
                                if (done) {
                                    if (fake) {
                                        uint4 o;
                                        __builtin_memcpy(&o, of, 16);
                                        *reinterpret_cast<uint4*>((T*)out + i * V) = o;
                                    } else {
                                        *reinterpret_cast<uint2*>((uint8_t*)out + i * V) = ob;
                                    }
                                }
                            }
                        }
                    }
                    if (!done) todo |= 1u << u;
                }
            }
            // one copy of the general encoder, rolled over pairs (inlined per vector and element it made the kernel 10.6 k
            // instructions long)
#pragma unroll 1
            for (int u = 0; u < U; ++u) {
                if (!((todo >> u) & 1u)) continue;
                uint4 r = raw[0];
#pragma unroll
                for (int v = 1; v < U; ++v)
                    if (u == v) r = raw[v];
                const int64_t i = i0 + u * stride;
                const float s = scale_of(i);
#pragma unroll 1
                for (int pr = 0; pr < V / 2; ++pr) {                    // a pair of elements at a time, one encoder instance
                    float w0, w1;
                    if constexpr (V == 8) {
                        uint32_t word = r.x;
                        if (pr == 1) word = r.y;
                        if (pr == 2) word = r.z;
                        if (pr == 3) word = r.w;
                        T a, b;
                        a.u = (uint16_t)(word & 0xffffu);
                        b.u = (uint16_t)(word >> 16);
                        w0 = to_f32<T>(a);
                        w1 = to_f32<T>(b);
                    } else {
                        w0 = __uint_as_float(pr ? r.z : r.x);
                        w1 = __uint_as_float(pr ? r.w : r.y);
                    }
                    T of[2];
                    uint8_t ob[2];
                    fp8_two<T>(w0, w1, s, tdt, mode & ~(FP8_EXACT_DIV | FP8_NO_PACKED16), of, ob);
                    const int64_t e = i * V + 2 * pr;
                    if (fake) {
                        ((T*)out)[e] = of[0];
                        ((T*)out)[e + 1] = of[1];
                    } else {
                        ((uint8_t*)out)[e] = ob[0];
                        ((uint8_t*)out)[e + 1] = ob[1];
                    }
                }
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < total; i += (int64_t)gridDim.x * FB) {
        const int64_t row = i / g;
        float s;
        if (static_scales) {
            s = load_as_f32(scales, row, sdt);
        } else {
            s = rnd(to_f32<T>(amax[row]) / fmax, sdt);
            if (s == 0.0f) s = 1.0f;
            if (i == row * g) store_from_f32(scales, row, sdt, s);
        }
        if (s == 0.0f) s = 1.0f;
        T of;
        uint8_t ob;
        fp8_one<T>(to_f32<T>(W[i]), s, tdt, mode & ~(FP8_EXACT_DIV | FP8_NO_PACKED16), DT, &of, &ob);
        if (fake) ((T*)out)[i] = of; else ((uint8_t*)out)[i] = ob;
    }
}

// AutoAWQ: one thread per output word
template <typename T>
__global__ __launch_bounds__(FB) void k_pack_awq_w(const T* __restrict__ W, const void* __restrict__ scales, int sdt,
                                                   const int32_t* __restrict__ zeros, int64_t R, int64_t K,
                                                   int64_t g, int32_t* __restrict__ qweight) {
    constexpr int WDT = dt_of<T>::value;
    const int p = promote(WDT, LLMC_F16);
    const int64_t R8 = R / 8, ng = K / g;
    const int64_t total = K * R8;
    const int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < total; i += (int64_t)gridDim.x * FB) {
        const int64_t rw = i / K, k = i - rw * K;   // k fastest: coalesced reads of W rows
        const int64_t gi = k / g;
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = rw * 8 + order[j];
            const float s = rnd(load_as_f32(scales, r * ng + gi, sdt), LLMC_F16);   // scales.to(float16)
            const float sz = rnd((float)zeros[r * ng + gi] * s, LLMC_F16);           // zeros * scales
            float t = rnd(to_f32<T>(W[r * K + k]) + sz, p);
            t = rnd(t / s, p);
            const int code = (int)rintf(t);
            word |= ((uint32_t)code) << (4 * j);
        }
        qweight[k * R8 + rw] = (int32_t)word;
    }
}
__global__ __launch_bounds__(FB) void k_pack_awq_z(const void* __restrict__ scales, int sdt,
                                                   const int32_t* __restrict__ zeros, int64_t R, int64_t ng,
                                                   int32_t* __restrict__ qzeros, uint16_t* __restrict__ sout) {
    const int64_t R8 = R / 8;
    const int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    for (int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x; i < ng * R; i += (int64_t)gridDim.x * FB) {
        const int64_t gi = i / R, r = i - gi * R;
        sout[gi * R + r] = f32_to_f16_bits(load_as_f32(scales, r * ng + gi, sdt));
        if (r < R8) {
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) word |= ((uint32_t)zeros[(r * 8 + order[j]) * ng + gi]) << (4 * j);
            qzeros[gi * R8 + r] = (int32_t)word;
        }
    }
}

static inline int grid_fb(int64_t n) {
    int64_t b = ceil_div64(n, FB);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

static inline int grid_cast(int64_t n) {
    int64_t b = ceil_div64(n, FB);
    return (int)(b > FP8_GRID_CAP ? FP8_GRID_CAP : (b < 1 ? 1 : b));
}

}  // namespace llmc

using namespace llmc;

extern "C" size_t llmc_fp8_quant_ws_bytes(int64_t G, int64_t g) {
    if (G <= 0 || g <= 0) return 0;
    return (((size_t)G * 4 + 255) & ~(size_t)255) + llmc_minmax_qparams_ws_bytes(G, g);
}

extern "C" int llmc_fp8_quant(const void* W, int dt, int64_t G, int64_t g, int fake, void* out, void* scales,
                              int sdt, int static_scales, void* ws, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(dt) && dtype_ok(sdt) && W && out && scales && G > 0 && g > 0, "fp8_quant: bad argument");
    LLMC_REQUIRE(static_scales || ws, "fp8_quant: workspace required for dynamic scales");
    void* amax = ws;
    if (!static_scales) {
        void* ws2 = (char*)ws + (((size_t)G * 4 + 255) & ~(size_t)255);
        // clamp(absmax, 1e-5) in dt: the symmetric qparams with qmax = 1
        int rc = llmc_minmax_qparams(W, dt, G, g, /*sym*/ 1, 1, -1.0f, 1.0f, amax, nullptr, ws2, stream);
        if (rc) return rc;
    }
    if (opt(OPT_FP8_EXACT_DIV)) fake |= FP8_EXACT_DIV;      // A/B switch, same results (include/llmc_hip.h)
    if (opt(OPT_FP8_NO_PACKED16)) fake |= FP8_NO_PACKED16;
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case LLMC_F16:
            hipLaunchKernelGGL((k_fp8_cast<f16_t>), dim3(grid_cast(G * g / (8 * FP8_U) + 1)), dim3(FB), 0, st, (const f16_t*)W,
                               (const f16_t*)amax, sdt, scales, static_scales, G, g, fake, out);
            break;
        case LLMC_BF16:
            hipLaunchKernelGGL((k_fp8_cast<bf16_t>), dim3(grid_cast(G * g / (8 * FP8_U) + 1)), dim3(FB), 0, st, (const bf16_t*)W,
                               (const bf16_t*)amax, sdt, scales, static_scales, G, g, fake, out);
            break;
        default:
            hipLaunchKernelGGL((k_fp8_cast<float>), dim3(grid_cast(G * g / (4 * FP8_U) + 1)), dim3(FB), 0, st, (const float*)W,
                               (const float*)amax, sdt, scales, static_scales, G, g, fake, out);
    }
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}

extern "C" int llmc_pack_awq_gemm(const void* weight, int wdt, const void* scales, int sdt, const int32_t* zeros,
                                  int64_t R, int64_t K, int64_t g, int32_t* qweight, int32_t* qzeros,
                                  void* scales_out_f16, llmc_stream_t stream) {
    LLMC_REQUIRE(dtype_ok(wdt) && dtype_ok(sdt), "pack_awq_gemm: bad dtype");
    LLMC_REQUIRE(weight && scales && zeros && qweight && qzeros && scales_out_f16 && R > 0 && K > 0 && g > 0,
                 "pack_awq_gemm: null/empty argument (AutoAWQ needs asymmetric zeros)");
    LLMC_REQUIRE(R % 8 == 0 && K % g == 0, "pack_awq_gemm: R must be a multiple of 8 and K of the group size");
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_fb(K * (R / 8));
    switch (wdt) {
        case LLMC_F16:
            hipLaunchKernelGGL((k_pack_awq_w<f16_t>), dim3(grid), dim3(FB), 0, st, (const f16_t*)weight, scales, sdt,
                               zeros, R, K, g, qweight);
            break;
        case LLMC_BF16:
            hipLaunchKernelGGL((k_pack_awq_w<bf16_t>), dim3(grid), dim3(FB), 0, st, (const bf16_t*)weight, scales, sdt,
                               zeros, R, K, g, qweight);
            break;
        default:
            hipLaunchKernelGGL((k_pack_awq_w<float>), dim3(grid), dim3(FB), 0, st, (const float*)weight, scales, sdt,
                               zeros, R, K, g, qweight);
    }
    LLMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_awq_z, dim3(grid_fb((K / g) * R)), dim3(FB), 0, st, scales, sdt, zeros, R, K / g, qzeros,
                       (uint16_t*)scales_out_f16);
    LLMC_LAUNCH_CHECK();
    return LLMC_OK;
}
