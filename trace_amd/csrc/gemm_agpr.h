// Accumulators in AGPRs that only inline asm names (gemm_w4.hip).
#pragma once
#include "common.h"

// The 64 accumulators of a wave live in a[4n : 4n + 3] (n = 8 * m-tile + n-tile) and the compiler never sees them as values: the MFMAs name their
// AGPRs in the asm text (clobbers keep hipcc out of them and size the kernel's AGPR allocation), the epilogue fetches them with v_accvgpr_read in asm.
// As C++ values ("a" operands, then "{a[..]}" operands) the register allocator — 256 of 256 AGPRs live, no slack — renumbered accumulators between the
// code paths of the K loop and reconciled the numberings with v_accvgpr copies and scratch: 200-600 VALU instructions and a dozen spills per 64 MFMAs.
// What that leaves to check in the ISA of every build: no v_accvgpr_write anywhere (hipcc parks spilled VGPRs in AGPRs it believes free).
#define W4_ACC_LIST(X) \
    X(0, 0, 1, 2, 3) X(1, 4, 5, 6, 7) X(2, 8, 9, 10, 11) X(3, 12, 13, 14, 15) X(4, 16, 17, 18, 19) X(5, 20, 21, 22, 23) X(6, 24, 25, 26, 27) X(7, 28, 29, 30, 31) X(8, 32, 33, 34, 35) X(9, 36, 37, 38, 39) X(10, 40, 41, 42, 43) X(11, 44, 45, 46, 47) X(12, 48, 49, 50, 51) X(13, 52, 53, 54, 55) X(14, 56, 57, 58, 59) X(15, 60, 61, 62, 63) X(16, 64, 65, 66, 67) X(17, 68, 69, 70, 71) X(18, 72, 73, 74, 75) X(19, 76, 77, 78, 79) X(20, 80, 81, 82, 83) X(21, 84, 85, 86, 87) X(22, 88, 89, 90, 91) X(23, 92, 93, 94, 95) X(24, 96, 97, 98, 99) X(25, 100, 101, 102, 103) X(26, 104, 105, 106, 107) X(27, 108, 109, 110, 111) X(28, 112, 113, 114, 115) X(29, 116, 117, 118, 119) X(30, 120, 121, 122, 123) X(31, 124, 125, 126, 127) X(32, 128, 129, 130, 131) X(33, 132, 133, 134, 135) X(34, 136, 137, 138, 139) X(35, 140, 141, 142, 143) X(36, 144, 145, 146, 147) X(37, 148, 149, 150, 151) X(38, 152, 153, 154, 155) X(39, 156, 157, 158, 159) X(40, 160, 161, 162, 163) X(41, 164, 165, 166, 167) X(42, 168, 169, 170, 171) X(43, 172, 173, 174, 175) X(44, 176, 177, 178, 179) X(45, 180, 181, 182, 183) X(46, 184, 185, 186, 187) X(47, 188, 189, 190, 191) X(48, 192, 193, 194, 195) X(49, 196, 197, 198, 199) X(50, 200, 201, 202, 203) X(51, 204, 205, 206, 207) X(52, 208, 209, 210, 211) X(53, 212, 213, 214, 215) X(54, 216, 217, 218, 219) X(55, 220, 221, 222, 223) X(56, 224, 225, 226, 227) X(57, 228, 229, 230, 231) X(58, 232, 233, 234, 235) X(59, 236, 237, 238, 239) X(60, 240, 241, 242, 243) X(61, 244, 245, 246, 247) X(62, 248, 249, 250, 251) X(63, 252, 253, 254, 255)
#define X(N, R0, R1, R2, R3)                                                                                                                \
    __device__ __forceinline__ void mfma_acc_##N(const bf16x8_t& w, const bf16x8_t& a) {                                                    \
        asm volatile("v_mfma_f32_16x16x32_" TRACE_EL " a[" #R0 ":" #R3 "], %0, %1, a[" #R0 ":" #R3 "]" :: "v"(w), "v"(a)                    \
                     : "a" #R0, "a" #R1, "a" #R2, "a" #R3);                                                                                 \
    }                                                                                                                                       \
    __device__ __forceinline__ void mfma_new_##N(const bf16x8_t& w, const bf16x8_t& a) {                                                    \
        asm volatile("v_mfma_f32_16x16x32_" TRACE_EL " a[" #R0 ":" #R3 "], %0, %1, 0" :: "v"(w), "v"(a) : "a" #R0, "a" #R1, "a" #R2, "a" #R3); \
    }                                                                                                                                       \
    __device__ __forceinline__ f32x4_t acc_get_##N() {                                                                                      \
        f32x4_t v;                                                                                                                          \
        asm volatile("v_accvgpr_read_b32 %0, a" #R0 "\n\tv_accvgpr_read_b32 %1, a" #R1 "\n\tv_accvgpr_read_b32 %2, a" #R2 "\n\tv_accvgpr_read_b32 %3, a" #R3 \
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) :: "a" #R0, "a" #R1, "a" #R2, "a" #R3);    /* (clobbers: hipcc must not park anything there around the read either) */ \
        return v;                                                                                                                           \
    }
W4_ACC_LIST(X)
#undef X
// the same with the accumulator index as a template argument (static_for: no switch for the optimiser to fold; past ~16 k IR instructions hipcc stops
// unrolling loops whose bodies carry a 64-way switch, the fragment arrays become scratch, and scratch accesses are VMEM operations counted waits do not know)
template <int N> struct AccReg;
#define X(N, R0, R1, R2, R3)                                                                                                                \
    template <> struct AccReg<N> {                                                                                                          \
        static __device__ __forceinline__ void acc(const bf16x8_t& w, const bf16x8_t& a) { mfma_acc_##N(w, a); }                            \
        static __device__ __forceinline__ void fresh(const bf16x8_t& w, const bf16x8_t& a) { mfma_new_##N(w, a); }                          \
        static __device__ __forceinline__ f32x4_t get() { return acc_get_##N(); }                                                           \
    };
W4_ACC_LIST(X)
#undef X

