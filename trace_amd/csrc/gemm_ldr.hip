// The 256x256 bf16 MFMA GEMM of the ViT / prefill projections with LOADER WAVES (the default for every shape gemm.hip's
// dispatcher gives 256^2 tiles; gemm.hip's own 256^2 kernel stays as variant 3 for A/B runs).
//
// Same tile, LDS image (XOR swizzle on the LDS-DMA source address), tile order, MFMA schedule and epilogue as gemm.hip — the
// results are bit-identical — but the workgroup is 8 MFMA waves (2 per SIMD) that never touch VMEM inside the K loop + 4 loader
// waves (1 per SIMD) that do nothing but issue the next K-tile's 64 LDS-DMA pieces (16 each) and wait for them.  Why: the CU's
// address path accepts one 1 KB global_load_lds piece per ~31 cycles, and a wave whose piece is waiting there cannot issue
// anything else; when that wave is also an MFMA wave (gemm.hip) the 64 pieces of a K-tile (~2000 cycles of address-path time)
// run in SERIES with its 64 MFMAs (knock-out runs in gemm.hip's header: 250 us + 243 us -> 458 us on fc1) instead of beside
// them.  With the stall moved to waves that have nothing else to do, measured on one MI355X (us, gemm.hip -> here):
// ViT fc1 98090x4096x1024 + QuickGELU 889 -> 736 (925 -> 1118 TFLOP/s), qkv x3072 640 -> 515 (965 -> 1198), prefill down
// 3934x4096x14336 394 -> 362 (1173 -> 1275).  12 waves per CU = 3 per SIMD = 168 registers per wave: the 128 accumulators + 24
// fragment registers of an MFMA wave just fit, the 16 prefetched residual pieces of gemm.hip's epilogue do not — they are
// requested after the accumulators have gone to LDS instead.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4;
constexpr int NMT = WM * WN * 64;                 // 512 MFMA threads
constexpr int NLW = 4;                            // loader waves
constexpr int NTHR = NMT + NLW * 64;              // 768
constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
constexpr int TOUCH_OFF = BM * (BN * 2 + 16) + (BM + BN) * 4;      // behind the staged output tile and the fp8 scale rows: 4 x 256 bytes of touch scratch

__device__ __forceinline__ int swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

typedef long long2_t __attribute__((ext_vector_type(2)));
template <bool FP8>
__device__ __forceinline__ f32x4_t mma_step(const bf16x8_t& w, const bf16x8_t& a, f32x4_t acc) {
    if constexpr (FP8) {      // fp8.hip's W8A8 path: a 16-byte fragment is two e4m3 MFMA operands (see gemm.hip)
        const long2_t w2 = __builtin_bit_cast(long2_t, w), a2 = __builtin_bit_cast(long2_t, a);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w2[0], a2[0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w2[1], a2[1], acc, 0, 0, 0);
    } else {
        return mfma16(w, a, acc);
    }
}

template <int EPI, bool FP8>
__global__ __launch_bounds__(NTHR) void gemm_ldr_kernel(GemmArgs p) {
    constexpr int ESZ = FP8 ? 1 : 2, CE = 16 / ESZ;
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    constexpr int OSTRIDE = BN * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    int t = xcd_remap(blockIdx.x, ntm * ntn);
    int tm, tn;
    {
        constexpr int GM = 8;
        const int per_group = GM * ntn;
        const int gid = t / per_group, first = gid * GM;
        const int gsz = min(ntm - first, GM);
        const int in_g = t - gid * per_group;
        tm = first + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K * ESZ / 128;                    // K-tiles of 128-byte rows

    if (wid >= WM * WN) {
        // ---------------- loader wave lw: pieces of 8 tile rows x 128 bytes; lw 0,1 -> A rows 0..127 / 128..255, lw 2,3 -> W ----------------
        const int lw = wid - WM * WN;
        const bool isA = lw < 2;
        const int half = (lw & 1) * 128;
        const char* src[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = half + j * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            src[j] = isA ? reinterpret_cast<const char*>(p.A) + ((size_t)min(m0 + row, p.M - 1) * p.lda + kc * CE) * ESZ
                         : reinterpret_cast<const char*>(p.W) + ((size_t)(n0 + row) * p.ldw + kc * CE) * ESZ;
        }
        const int region = (isA ? 0 : A_BYTES) + half * 128;
        auto issue = [&](int kt) {
            char* dst = smem + (kt & 1) * STAGE + region;
            const int ko = kt * 128;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + ko),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        };
        // L2 touches of the A rows, LEAD K-tiles ahead of their LDS-DMA pieces (gemm_pers.hip's loader has the full story: with two LDS stages one
        // K-tile is in flight, so a K-tile costs the load's latency unless the line is already in L2).  One instruction per workgroup and K-tile:
        // the 4 workgroups that share a row panel (column tiles tn, tn + 1, ..) touch 64 rows of it each.  p.opt bit 1 switches them off (A/B).
        constexpr int LEAD = 3;
        const bool atouch = lw == 0 && !(p.opt & 2);
        const char* tsrc = reinterpret_cast<const char*>(p.A) + (size_t)min(m0 + (tn & 3) * 64 + lane, p.M - 1) * p.lda * ESZ;
        // a touch = one dword per lane by LDS-DMA into the wave's 256-byte scratch behind the staged tile: an L2 fill with no register destination
        // (a dummy register written by an asm load is dead to the compiler at once — and overwritten for real microseconds later: gemm_pers.hip)
        char* tscr = smem + TOUCH_OFF + lw * 256;
#define LDR_TOUCH_AT(PTR)                                                                                                          \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTR), (__attribute__((address_space(3))) void*)tscr, 4, 0, 0)
#define LDR_TOUCH(KT) LDR_TOUCH_AT(tsrc + (KT) * 128)
        const bool rtouch = EPI == EPI_RESIDUAL && !FP8 && nk > 2 && !(p.opt & 1);
        int pend = 0;                              // touch loads issued behind the latest batch of pieces: the in-order counter is waited down to them
        issue(0);
        if (atouch) {
            if (1 < nk) { LDR_TOUCH(1); ++pend; }
            if (2 < nk) { LDR_TOUCH(2); ++pend; }
        }
        for (int kt = 0; kt < nk; ++kt) {
            // own pieces of tile kt landed (not the touches behind them); everyone is done with tile kt-1
            if (pend == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (pend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            pend = 0;
            if (kt + 1 < nk) issue(kt + 1);
            if (atouch && kt + LEAD < nk) { LDR_TOUCH(kt + LEAD); pend = 1; }
            if (rtouch && kt + 2 == nk) {
                // Residual touches: one byte of each of the tile's 1024 residual lines (256 rows x 512 bytes), four per loader lane, behind the
                // LAST K-tile's pieces, so that the MFMA waves' residual loads — requested only after the K loop, when the accumulators have left
                // their registers — come from L2 instead of waiting ~2 us for HBM with nothing left to overlap.  (One K-tile earlier, waited for
                // with everything else: 2-6 % SLOWER than none — tools/gemm_pers_ab.py, r02_gemm_pers_ab.txt.)
                const int L = lw * 64 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = L + q * 256, row = idx >> 2, seg = idx & 3;
                    const bf16_t* ra = p.R + (size_t)min(m0 + row, p.M - 1) * p.ldr + n0 + seg * 64;
                    LDR_TOUCH_AT(ra);
                }
                pend = 4;                          // (kt + LEAD >= nk here: no A touch in this iteration)
            }
        }
#undef LDR_TOUCH
#undef LDR_TOUCH_AT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA (touch) of this wave in flight beyond this point (LDS is released with the workgroup)
        __syncthreads();                           // the MFMA waves' epilogue barriers (one more on the fp8 path: the scale rows)
        if (FP8) __syncthreads();
        __syncthreads();
        return;
    }

    // ---------------- MFMA waves ----------------
    const int r = lane & 15, g = lane >> 4;
    const int wm = wid / WN, wn = wid % WN;
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int abase = swz(wm * (BM / WM) + r, g), wbase = swz(wn * (BN / WN) + r, g);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* fa = sa + (abase ^ (ks << 6));
            const char* fw = sw + (wbase ^ (ks << 6));
            bf16x8_t wf[TN], ac[2], an[2];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(fw + j * 2048);
            ac[0] = *reinterpret_cast<const bf16x8_t*>(fa);
            ac[1] = *reinterpret_cast<const bf16x8_t*>(fa + 2048);
            __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#pragma unroll
            for (int ip = 0; ip < TM / 2; ++ip) {
                if constexpr (FP8) {
                    // fp8: 16 MFMAs per pair of m-tiles instead of 8, and no room for the second fragment pair beside 128 accumulators
                    // (the prefetch spilled inside the loop): the next pair's first fragment is prefetched under this pair's MFMAs, its
                    // second one is read behind the first eight
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[2 * ip][j] = mma_step<true>(wf[j], ac[0], acc[2 * ip][j]);
                    if (ip + 1 < TM / 2) ac[0] = *reinterpret_cast<const bf16x8_t*>(fa + (2 * ip + 2) * 2048);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[2 * ip + 1][j] = mma_step<true>(wf[j], ac[1], acc[2 * ip + 1][j]);
                    if (ip + 1 < TM / 2) ac[1] = *reinterpret_cast<const bf16x8_t*>(fa + (2 * ip + 3) * 2048);
                } else {
                    if (ip + 1 < TM / 2) {
                        an[0] = *reinterpret_cast<const bf16x8_t*>(fa + (2 * ip + 2) * 2048);
                        an[1] = *reinterpret_cast<const bf16x8_t*>(fa + (2 * ip + 3) * 2048);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[2 * ip + ii][j] = mma_step<false>(wf[j], ac[ii], acc[2 * ip + ii][j]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                    if (ip + 1 < TM / 2) { ac[0] = an[0]; ac[1] = an[1]; }
                }
            }
        }
    }

    constexpr int OUTW = GLU ? BN / 2 : BN;
    constexpr int CPR = OUTW / 8;
    const int on0 = GLU ? n0 / 2 : n0;
    constexpr int OIT = BM * CPR / NMT;
    float bv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        uint2 b2 = make_uint2(0u, 0u);
        if (!FP8 && !GLU && p.bias) b2 = *reinterpret_cast<const uint2*>(p.bias + n0 + wn * (BN / WN) + j * 16 + g * 4);   // (the fp8 path is the bias-free LLM)
        bv[j][0] = bflo(b2.x); bv[j][1] = bfhi(b2.x); bv[j][2] = bflo(b2.y); bv[j][3] = bfhi(b2.y);
    }
    // fp8: the 256 row scales of A and the 256 row scales of W go through LDS (behind the staged tile): holding a lane's 8 + 16 of
    // them in registers beside the 128 accumulators spilled
    float* s_sa = reinterpret_cast<float*>(smem + BM * OSTRIDE);
    float* s_sw = s_sa + BM;
    __syncthreads();
    if constexpr (FP8) {
        if (tid < BM) s_sa[tid] = p.sa[min(m0 + tid, p.M - 1)];
        else if (tid < BM + BN) s_sw[tid - BM] = p.sw[n0 + tid - BM];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = wm * (BM / WM) + i * 16 + r;
        if (!GLU) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * (BN / WN) + j * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = acc[i][j][q];
                    if constexpr (FP8) x *= s_sa[mrow] * s_sw[nl + q];
                    else x += bv[j][q];
                    if (EPI == EPI_QUICKGELU) x = x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
                    v[q] = x;
                }
                *reinterpret_cast<uint2*>(smem + mrow * OSTRIDE + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < TN / 2; ++jj) {
                const int nl = wn * (BN / WN / 2) + jj * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float gt = acc[i][2 * jj][q], up = acc[i][2 * jj + 1][q];
                    if constexpr (FP8) {
                        const int cg = wn * (BN / WN) + (2 * jj) * 16 + g * 4 + q;       // gate column of the tile; its up column is 16 further
                        gt *= s_sa[mrow] * s_sw[cg]; up *= s_sa[mrow] * s_sw[cg + 16];
                    }
                    v[q] = gt * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;
                }
                *reinterpret_cast<uint2*>(smem + mrow * OSTRIDE + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
    }
    // residual pieces: requested once the accumulators are in LDS (with 168 registers per wave there is no room for them beside
    // the accumulators, as gemm.hip has), all in flight across the barrier
    uint4 rres[EPI == EPI_RESIDUAL ? OIT : 1];
    if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int it = 0; it < OIT; ++it) {
            const int c = it * NMT + tid, row = c / CPR, ch = c - row * CPR;
            const int m = min(m0 + row, p.M - 1);
            rres[it] = *reinterpret_cast<const uint4*>(p.R + (size_t)m * p.ldr + on0 + ch * 8);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < OIT; ++it) {
        const int c = it * NMT + tid;
        const int row = c / CPR, ch = c - row * CPR;
        const int m = m0 + row;
        if (m >= p.M) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * OSTRIDE + ch * 16);
        if (EPI == EPI_RESIDUAL) {
            const uint4 rr = rres[it];
            v.x = pack2bf(bflo(v.x) + bflo(rr.x), bfhi(v.x) + bfhi(rr.x));
            v.y = pack2bf(bflo(v.y) + bflo(rr.y), bfhi(v.y) + bfhi(rr.y));
            v.z = pack2bf(bflo(v.z) + bflo(rr.z), bfhi(v.z) + bfhi(rr.z));
            v.w = pack2bf(bflo(v.w) + bflo(rr.w), bfhi(v.w) + bfhi(rr.w));
        }
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + on0 + ch * 8));
    }
}

template <int EPI, bool FP8>
void launch_one(const GemmArgs& p, int nblk, size_t lds, hipStream_t s) {
    static LdsGrant grant;
    (void)grant_dynamic_lds(grant, reinterpret_cast<const void*>(gemm_ldr_kernel<EPI, FP8>), (int)lds);       // a refusal shows as the launch error the caller checks
    hipLaunchKernelGGL((gemm_ldr_kernel<EPI, FP8>), dim3(nblk), dim3(NTHR), lds, s, p);
}

}  // namespace

int g_gemm_ldr_opt = 0;            // A/B: bit 0 = no residual touches, bit 1 = no A-panel touches (trace_op_set_gemm_variant(400 + opt))
int launch_gemm_ldr(const GemmArgs& p0, int epi, hipStream_t s) {
    GemmArgs p = p0;
    p.opt = g_gemm_ldr_opt;
    if (p.M < 1 || p.N % BN || p.K % BK) return TRACE_ERR_ARG;
    constexpr size_t LOOPB = 2 * STAGE, OBYTES = (size_t)TOUCH_OFF + 4 * 256;      // staged tile + the fp8 scale rows + the touch scratch
    const size_t lds = LOOPB > OBYTES ? LOOPB : OBYTES;
    const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
    if (p.fp8) {
        switch (epi) {
            case EPI_NONE: launch_one<EPI_NONE, true>(p, nblk, lds, s); break;
            case EPI_RESIDUAL: launch_one<EPI_RESIDUAL, true>(p, nblk, lds, s); break;
            case EPI_SWIGLU: launch_one<EPI_SWIGLU, true>(p, nblk, lds, s); break;
            default: return TRACE_ERR_ARG;
        }
        return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
    }
    switch (epi) {
        case EPI_NONE: launch_one<EPI_NONE, false>(p, nblk, lds, s); break;
        case EPI_RESIDUAL: launch_one<EPI_RESIDUAL, false>(p, nblk, lds, s); break;
        case EPI_QUICKGELU: launch_one<EPI_QUICKGELU, false>(p, nblk, lds, s); break;
        case EPI_SWIGLU: launch_one<EPI_SWIGLU, false>(p, nblk, lds, s); break;
        default: return TRACE_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
