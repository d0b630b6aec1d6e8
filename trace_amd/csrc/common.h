// Shared device helpers for the TRACE gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA 16x16x32 /
// 32x32x16 (bf16, or fp16 in the TRACE_F16 build) with fp32 accumulators, LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit element type of a build: bf16 by default; IEEE half when compiled with -DTRACE_F16 (libtrace_hip_f16.so, the reference's own fp16
// inference dtype: trace/model/builder.py:50,127,147).  Everything that MOVES elements — LDS-DMA, tile layouts, fragment reads, the KV cache — only
// knows "16 bits"; the element type enters through the helpers below (conversions, the two MFMA shapes, the asm mnemonic suffix, packed 1.0).
// The names keep "bf" for both builds: bf16_t is "the build's 16-bit element", pack2bf "two fp32 -> one packed pair, round to nearest even".
typedef uint16_t bf16_t;                                                   // raw element bits in HBM
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;                // MFMA A/B fragment (8 elements = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;                 // 16x16 MFMA C/D fragment
typedef __attribute__((ext_vector_type(16))) float f32x16_t;               // 32x32 MFMA C/D fragment

#ifdef TRACE_F16
#define TRACE_ELEMENT_TYPE 1
#define TRACE_EL "f16"                                                      // instruction-name suffix for inline asm
#define TRACE_EL_ONE2 0x3c003c00u                                           // (1.0, 1.0) packed
typedef __attribute__((ext_vector_type(2))) _Float16 el2_native_t;
typedef __attribute__((ext_vector_type(8))) _Float16 el8_native_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float bflo(uint32_t v) { return (float)__builtin_bit_cast(el2_native_t, v)[0]; }   // low half of a packed pair
__device__ __forceinline__ float bfhi(uint32_t v) { return (float)__builtin_bit_cast(el2_native_t, v)[1]; }   // high half
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {             // round to nearest even, overflow -> inf (as torch.float16)
    const el2_native_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ f32x4_t mfma16(const bf16x8_t& w, const bf16x8_t& a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(el8_native_t, w), __builtin_bit_cast(el8_native_t, a), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma32(const bf16x8_t& w, const bf16x8_t& a, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(el8_native_t, w), __builtin_bit_cast(el8_native_t, a), c, 0, 0, 0);
}
__device__ __forceinline__ float dot2_el(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(el2_native_t, a), __builtin_bit_cast(el2_native_t, b), c, false);
}
#else
#define TRACE_ELEMENT_TYPE 0
#define TRACE_EL "bf16"
#define TRACE_EL_ONE2 0x3f803f80u
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bflo(uint32_t v) { return __uint_as_float(v << 16); }          // low half of a packed pair
__device__ __forceinline__ float bfhi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }  // high half
// fp32 -> bf16, round-to-nearest-even: the native __bf16 conversion lowers to ONE v_cvt_pk_bf16_f32 on gfx950
// (a hand-rolled integer rounding sequence costs ~16 VALU ops per pair and dominated epilogues / softmax packing).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2_native_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ f32x4_t mfma16(const bf16x8_t& w, const bf16x8_t& a, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t mfma32(const bf16x8_t& w, const bf16x8_t& a, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0); }
__device__ __forceinline__ float dot2_el(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_native_t, a), __builtin_bit_cast(bf16x2_native_t, b), c, false);
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the 64 lanes on DPP (VALU speed, no LDS crossbar round trips): quad swaps, row mirrors, then the two row broadcasts leave the
// total in lane 63; readlane returns it to every lane (as a scalar).  Fixed order.  __shfl_xor compiles to ds_bpermute_b32 — ~120
// cycles of latency per step, which made per-row reductions the critical path of the slot pool.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define TRACE_DPP_ADD(CTRL, ROWMASK) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, true))
    TRACE_DPP_ADD(0xB1, 0xf);      // quad_perm [1,0,3,2]
    TRACE_DPP_ADD(0x4E, 0xf);      // quad_perm [2,3,0,1]
    TRACE_DPP_ADD(0x141, 0xf);     // row_half_mirror
    TRACE_DPP_ADD(0x140, 0xf);     // row_mirror: every lane of a 16-lane row holds the row's sum
    TRACE_DPP_ADD(0x142, 0xa);     // row_bcast15 into rows 1, 3
    TRACE_DPP_ADD(0x143, 0xc);     // row_bcast31 into rows 2, 3: lane 63 holds the wave's sum
#undef TRACE_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: the dispatcher places block b on XCD b%8, so give each
// XCD a contiguous chunk of logical tile ids (neighbouring tiles share operand panels in that XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

#define TRACE_OK 0
#define TRACE_ERR_ARG (-1)
#define TRACE_ERR_HIP (-2)
#define TRACE_ERR_STATE (-3)
