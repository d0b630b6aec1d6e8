// What the two persistent 256x256 GEMM kernels (gemm_pers.hip: 8 MFMA + 4 loader waves; gemm_w4.hip: 4 waves of 128x128) share: the tile order, each
// workgroup's share of it, the per-XCD ticket counters' layout, and the lane transposition of their register epilogues.
#pragma once
#include "common.h"
#include "kernels.h"

namespace tilewalk {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int CTR_STRIDE = 32;                    // ints between the per-XCD ticket counters (one 128-byte line each)

// logical tile id -> (row panel, column panel): groups of 8 row panels x all column panels (gemm.hip's order)
__device__ __forceinline__ void tile_coords(int t, int ntm, int ntn, int& tm, int& tn) {
    constexpr int GM = 8;
    const int per_group = GM * ntn;
    const int gid = t / per_group, first = gid * GM;
    const int gsz = min(ntm - first, GM);
    const int in_g = t - gid * per_group;
    tm = first + in_g % gsz;
    tn = in_g / gsz;
}

// which tiles this workgroup walks: its XCD's contiguous chunk of logical tile ids (common.h xcd_remap's split), first tile = its slot
struct Sched {
    int ntm, ntn, nk, xcd, slot, cnt, base, nwg;
};
__device__ __forceinline__ Sched make_sched(const GemmArgs& p) {
    Sched sc;
    sc.ntn = p.N / BN; sc.ntm = (p.M + BM - 1) / BM; sc.nk = p.K / BK;
    const int total = sc.ntm * sc.ntn, G = gridDim.x;
    sc.xcd = blockIdx.x & 7; sc.slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    sc.cnt = q8 + (sc.xcd < r8 ? 1 : 0);
    sc.base = (sc.xcd < r8) ? sc.xcd * (q8 + 1) : r8 * (q8 + 1) + (sc.xcd - r8) * q8;
    sc.nwg = (G >> 3) + (sc.xcd < (G & 7) ? 1 : 0);                // workgroups on this XCD (<= cnt: the launcher keeps G <= total)
    return sc;
}

// v_permlane16_swap: rows (16 lanes) 1 and 3 of a <-> rows 0 and 2 of b
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    const u2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

}  // namespace tilewalk
