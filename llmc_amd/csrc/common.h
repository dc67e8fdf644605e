// common.h — shared device/host helpers for libllmc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/llmc_hip.h"

namespace llmc {

// ---- error plumbing (no exception crosses the ABI) -------------------------------------------
void set_last_error(const char* where, hipError_t e);
void set_last_error_msg(const char* msg);

#define LLMC_HIP_CHECK(expr)                                   \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) {                                \
            ::llmc::set_last_error(#expr, _e);                 \
            return LLMC_EIO;                                   \
        }                                                      \
    } while (0)

#define LLMC_LAUNCH_CHECK()                                    \
    do {                                                       \
        hipError_t _e = hipGetLastError();                     \
        if (_e != hipSuccess) {                                \
            ::llmc::set_last_error("kernel launch", _e);       \
            return LLMC_EIO;                                   \
        }                                                      \
    } while (0)

#define LLMC_REQUIRE(cond, msg)                                \
    do {                                                       \
        if (!(cond)) {                                         \
            ::llmc::set_last_error_msg(msg);                   \
            return LLMC_EINVAL;                                \
        }                                                      \
    } while (0)

// per-device facts / attributes (abi.hip; mutex-guarded, safe from several host threads)
static constexpr int LLMC_MAX_DEVICES = 64;
int device_cu_count();                                  // CUs of the current device (256 on MI355X)
int ensure_dynamic_lds(const void* fn, int bytes);
int cu_reserve();                                       // llmc_hip_set_cu_reserve (per calling thread)
bool helper_streams_enabled();                          // llmc_hip_set_helper_streams (per calling thread)

// Explicit A/B switches (llmc_hip_set_option; per calling thread, all 0 by default, every value produces valid — and, unless
// the table in include/llmc_hip.h says otherwise, bit-identical — results). The library reads NO environment variable.
enum Opt : int {
    OPT_K3_FP32 = 0,        // K3's large products on the fp32 MFMA pipe instead of split-bf16
    OPT_K3_NO_GEMM6,        // deep levels of the triangular inverse on k_gemm3 instead of gemm6
    OPT_K3_NO_PLANES,       // K3's far updates on k_gemm3 (split in every tile) instead of pre-split planes + k_gemm3s
    OPT_K3_SPLIT_FAR,       // K3 far update of an outer block as two launches instead of one
    OPT_K4_SPLIT_FAR,       // K4 far update of a column group as three launches instead of one
    OPT_K4_ERR_ROWMAJOR,    // K4 error columns row-major instead of k-major
    OPT_GPTQ_GENERIC,       // k_gptq_block: generic IEEE-division path for every wave
    OPT_GEMM3_NOSPEC,       // never k_gemm3s
    OPT_GEMM3S_MIN_TILES,   // tile-count threshold of k_gemm3s (0 = the built-in 48 / 256)
    OPT_NO_SHORTK,          // short products through the general fp32 GEMM
    OPT_LINEAR_NOSPLIT,     // k-tiled GEMM never cuts a small product into k-slices
    OPT_FP8_EXACT_DIV,      // FP8 cast through the IEEE division + general encoder
    OPT_SIDE_CU_MASK,       // helper streams created with a CU mask (read when a caller stream's helper set is first created)
    OPT_K1_BATCH_OFF,       // llmc_hessian_accum_multi as one launch per problem instead of one tile queue
    OPT_K1_FP32_DIAG,
    OPT_FP8_NO_PACKED16,    // FP8 cast: the float form of the division-free path instead of the packed 16-bit one
    OPT_GEMM3S_NO_DMA,      // k_gemm3s planes form: producers copy through registers + ds_write (round 5) instead of LDS-DMA
    OPT_SGEMM_NO_WIDE,      // K4's phased far update on k_sgemm (128 x 128 tiles, two workgroups per CU) instead of k_sgemm_wide
    OPT_GEMM3_NO_WIDE,      // K3's far updates on k_gemm3s (one 512-thread workgroup per CU) instead of k_gemm3w (two 128 x 128 workgroups)
    OPT_COUNT
};
int opt(int id);
#ifdef LLMC_LAB
// lab builds only (tools/probes; never in the shipped library): ablation / debug switches read from the environment
static inline const char* lab_env(const char* k) { return getenv(k); }
#endif

static inline int dtype_size(int dt) { return dt == LLMC_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == LLMC_F16 || dt == LLMC_BF16 || dt == LLMC_F32; }

// ---- storage types ------------------------------------------------------------------------------
struct f16_t { uint16_t u; };
struct bf16_t { uint16_t u; };

template <typename T> struct dt_of;
template <> struct dt_of<f16_t> { static constexpr int value = LLMC_F16; };
template <> struct dt_of<bf16_t> { static constexpr int value = LLMC_BF16; };
template <> struct dt_of<float> { static constexpr int value = LLMC_F32; };

__device__ __forceinline__ float f16_bits_to_f32(uint16_t u) {
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float v) {
    // ATen evaluates an fp16 op in fp32 and THEN converts: two roundings. hipcc would fuse `(f16)(a * b)` / `(f16)(a + b)`
    // into v_fma_mixlo_f16, which rounds the exact result once — different on the rare values where the fp32 rounding
    // lands on an fp16 tie (round 3: `org_max * 0.95f` in the AWQ clip search chose another clip level than the reference
    // for 1-2 % of the groups). The barrier materialises the fp32 value first. (bf16 has no fused form.)
    asm volatile("" : "+v"(v));
    _Float16 h = (_Float16)v;  // RNE
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t u) {
    return __uint_as_float(((uint32_t)u) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float v) {
    // gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN stays a quiet NaN); the integer
    // emulation this replaces (NaN test, lsb, add, shift) was ~6 VALU ops on every rounding of the ATen-style chains
    const __bf16 h = (__bf16)v;
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// Make an fp32 value opaque to the optimiser. hipcc folds `(f16)(a * b)` into v_fma_mixlo_f16 a, b, +0: the added +0
// turns a product of -0.0 into +0.0 ((-0) + (+0) = +0), which the reference's `(q - 0) * s` keeps negative.
__device__ __forceinline__ float opaque_f32(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

// round an fp32 value to storage dtype `dt` and come back to fp32 (ATen "opmath" semantics)
__device__ __forceinline__ float rnd(float v, int dt) {
    if (dt == LLMC_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
    if (dt == LLMC_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(v));
    return v;
}
template <int DT> __device__ __forceinline__ float rndc(float v) {
    if constexpr (DT == LLMC_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
    else if constexpr (DT == LLMC_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(v));
    else return v;
}

__device__ __forceinline__ float load_as_f32(const void* p, int64_t i, int dt) {
    if (dt == LLMC_F16) return f16_bits_to_f32(((const uint16_t*)p)[i]);
    if (dt == LLMC_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    return ((const float*)p)[i];
}
__device__ __forceinline__ void store_from_f32(void* p, int64_t i, int dt, float v) {
    if (dt == LLMC_F16) ((uint16_t*)p)[i] = f32_to_f16_bits(v);
    else if (dt == LLMC_BF16) ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
    else ((float*)p)[i] = v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return f16_bits_to_f32(v.u); }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_bits_to_f32(v.u); }
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return f16_t{f32_to_f16_bits(v)}; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return bf16_t{f32_to_bf16_bits(v)}; }
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }

// torch promotion among the three float dtypes (f16 + bf16 -> f32)
__host__ __device__ __forceinline__ int promote(int a, int b) {
    if (a == b) return a;
    return LLMC_F32;
}

// 16-byte vector of T
template <typename T> struct vec16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

// wave64 helpers
__device__ __forceinline__ float wave_max(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int pow2_ceil(int64_t x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace llmc
