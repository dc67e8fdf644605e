// mfma_common.h — types and helpers shared by the 16-bit MFMA kernels (hessian_syrk.hip, linear_eval.hip).
#pragma once
#include "common.h"

namespace llmc {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define LDS_AS __attribute__((address_space(3)))

template <int DT> struct Mfma;
template <> struct Mfma<LLMC_BF16> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<LLMC_F16> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

typedef __attribute__((ext_vector_type(4))) int i32x4;

// One LDS-DMA piece: 64 lanes x 16 B from buffer(rsrc)+voff[lane] to LDS lds_addr + 16*lane.
// Issued from inline asm ON PURPOSE: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first
// ds_read that follows a compiler-visible LDS-DMA (it cannot prove the two do not alias), which would
// serialise the prefetch of K-step k+1 with the MFMAs of K-step k. The asm form is invisible to that
// pass; completion is waited for by hand (dma_wait_all) before the barrier that publishes the stage.
// M0 (LDS base of the DMA) is written in the same statement and restored; s_nop covers the
// SGPR->VMEM and M0->VMEM wait states that hipcc does not pad inside an asm string.
__device__ __forceinline__ void dma16(i32x4 rsrc, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_addr)
        : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


}  // namespace llmc
