// bf16 MFMA GEMM for the ViT and LLM-prefill projections:  C[M,N] = A[M,K] . W[N,K]^T (+ epilogue)
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear [out,in] layout), which is the natural
// MFMA layout: every lane's 8-element fragment is one 16-byte read.  16x16x32 bf16 MFMA, fp32 accumulate.
// Operands are fed swapped (D^T = W.A^T) so each lane ends up with 4 consecutive output columns of one row; the
// tile is then staged through LDS and leaves as full 16-byte row-contiguous stores (bias / QuickGELU / SwiGLU are
// applied in fp32 registers first, the residual is added at the coalesced stage).
// Two tile shapes: 256x256 (1 workgroup/CU) for the big ViT / gate|up GEMMs — run by gemm_ldr.hip's loader-wave version of this
// file's kernel since the end of round 1 — and 128x128 (4 waves) when a 256-tiling would leave CUs idle (LLM prefill M ~ 2k with N = 4096).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ int swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

// =========================================================================================================
// Direct-to-LDS (global_load_lds, 16 B/lane) double-buffered tiles, BMxBNx64, (WM x WN) waves.
// The LDS image is lane-linear per wave instruction (HW: M0 base + lane*16), so the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane *source* address (same involution on
// the read side).  One barrier per K-tile: tile kt+1 streams into the other buffer while tile kt feeds the MFMAs.
// Tile order: XCD-contiguous, then grouped (8 row panels x all column panels) so the ~32 workgroups resident on
// one XCD share A row-panels and W column-panels through that XCD's L2.
// Measured dead ends (kept out of the code): (1) a deeper pipeline — A triple-buffered two tiles ahead with counted
// vmcnt + raw s_barrier — gave nothing: the loop is not load-latency bound; (2) a persistent one-workgroup-per-CU
// version that prefetches the next output tile's first K-tile under a two-half epilogue was 5-15 % SLOWER on the
// K = 1024 ViT shapes than letting the dispatcher place fresh workgroups (static tile assignment + two extra
// epilogue barriers cost more than the hidden prologue); (3) de-phasing the first wave of workgroups (groups of CUs
// sleeping 3-10 us before their first tile so that the tiles' HBM write bursts do not coincide) moved the ViT shapes by
// -4..+1 % — the epilogues are not synchronised enough for write bandwidth to be what the fixed cost pays for.
// (4) two co-resident workgroups per CU (256x128 tiles, 4 waves, 32-wide K-tiles in 64-byte swizzled rows, 70 KB of LDS
// each) so that one tile's prologue/epilogue overlaps the other's main loop: correct, but 730 vs 761 (qkv), 788 vs 808
// (fc1), 832 vs 986 (fc2) TFLOP/s — the halved work per barrier costs what the overlap buys.
// (5) an 8-wave ping-pong main loop (waves w / w+4 of a SIMD alternating a fragment-read + LDS-DMA phase with a 32-MFMA
// phase across raw s_barriers, set B one phase behind): bit-correct, same speed (885 vs 876, 897 vs 931 TFLOP/s).
// Knock-out runs on that loop (fc1, 690 us) explain why: MFMAs alone 250 us (= the 2.5 PFLOP/s rate), LDS-DMA alone
// 243 us (the CU's address path retires one 1 KB piece per ~31 cycles = 32 B/clk, so 64 KB per K-tile costs as much as
// its 64 MFMAs per wave), fragment reads alone 83 us, everything outside the K loop 205-232 us — and MFMA + DMA run
// almost additively (458 us together), in either loop structure, also with the DMA confined to the read phases.  A
// 256x256 tile cannot lower DMA bytes per flop (the accumulators already fill half the register file), so the next
// step is hiding the per-tile 8 us (epilogue 4 + stores 2 + prologue 1.6) rather than re-shaping the K loop.
// (6) operand panels read K-tile-contiguously (what a tiled producer layout would give; timing experiment on the same
// bytes): +2-4 % only — unlike the decode GEMV, these panels come from L2 / infinity cache, not DRAM pages.
// Per-tile cost after the epilogue fixes (tools/gemm_trace.py): ~8 us fixed + ~1.7 us per K-tile.
// (7) What did work: LOADER WAVES — gemm_ldr.hip is this kernel with the LDS-DMA issue moved to four extra waves (+20-24 % on
// the K = 1024 shapes, bit-identical); the reason (1)-(5) changed nothing is that in all of them the waves that wait on the
// address path are the waves that should be issuing MFMAs.
// FP8 = true (fp8.hip's W8A8 path): A and W are e4m3 bytes — the same 128-byte LDS rows then hold 128 k instead of 64, a lane's
// 16-byte fragment read is TWO fp8 MFMA operands (v_mfma_f32_16x16x32_fp8_fp8; which k a slot means only has to agree between the two
// operands), and the epilogue multiplies the accumulators by the row scale of A and the row scale of W.
typedef long long2_t __attribute__((ext_vector_type(2)));
template <bool FP8>
__device__ __forceinline__ f32x4_t mma_step(const bf16x8_t& w, const bf16x8_t& a, f32x4_t acc) {
    if constexpr (FP8) {
        const long2_t w2 = __builtin_bit_cast(long2_t, w), a2 = __builtin_bit_cast(long2_t, a);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w2[0], a2[0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w2[1], a2[1], acc, 0, 0, 0);
    } else {
        return mfma16(w, a, acc);
    }
}

// WT: the W operand comes from the decode tile copy [N/16][K/64][64 lanes][16] (GemmArgs::w_tiled): a 16-row x 64-k block is one contiguous 2 KB,
// lane-linear in MFMA-fragment order — lane (r, g) holds W[r][g*16 .. +16], its two fragments (k-steps 0 / 1 of the K-tile) back to back — so the
// LDS image of a K-tile is [tile][k-step][lane][16 B], read with plain lane-linear ds_read_b128, and the A fragments take the matching k-slots
// (16-byte chunk 2g + ks of the row instead of 4 ks + g: an MFMA only needs both operands to agree on which k a slot means).
// NSTAGE > 2 (the decode GEMMs: M <= 128 rows, weights streamed once from HBM, nothing to re-use): a ring of NSTAGE K-tiles with NSTAGE - 1 in
// flight behind counted vmcnt waits and one raw s_barrier per K-tile — the double buffer keeps ONE 16 KB weight tile per workgroup under way, and
// with a single workgroup on most CUs (224-256 of them) that is a few GB/s per CU of a stream that wants 25.
template <int BM, int BN, int WM, int WN, int EPI, bool FP8, bool WT = false, int NSTAGE = 2>
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_kernel(GemmArgs p) {
    static_assert(!WT || (!FP8 && BN % 16 == 0), "tiled weights: bf16 only");
    constexpr int ESZ = FP8 ? 1 : 2, CE = 16 / ESZ;       // element bytes, elements per 16-byte chunk
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int A_IT = BM * 8 / NTHR, W_IT = BN * 8 / NTHR;
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    constexpr int OSTRIDE = BN * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wid / WN, wn = wid % WN;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    // EPI_PARTIAL: blockIdx = chunk * tiles + tile — the tiles of one K-chunk are neighbours, so the 8 XCDs stream 8 different weight panels
    const int kchunk = EPI == EPI_PARTIAL ? blockIdx.x / (ntm * ntn) : 0;
    int t = xcd_remap(EPI == EPI_PARTIAL ? blockIdx.x - kchunk * (ntm * ntn) : blockIdx.x, ntm * ntn);
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
    // phase stamps (100 MHz s_memrealtime) of wave 0, only when a trace buffer is passed
    unsigned long long* tr = p.trace ? p.trace + (size_t)blockIdx.x * 8 : nullptr;
    if (tr && threadIdx.x == 0) { tr[0] = __builtin_amdgcn_s_memrealtime(); tr[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); tr[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }

    const char* asrc[A_IT];
    const char* wsrc[W_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int slot = i * NTHR + tid, row = slot >> 3, kc = (slot & 7) ^ ((row >> 1) & 7);
        const int am = min(m0 + row, p.M - 1);
        asrc[i] = reinterpret_cast<const char*>(p.A) + ((size_t)am * p.lda + kc * CE) * ESZ;
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        if constexpr (WT) {
            const int q = i * NW + wid;                    // 1 KB piece (16-row tile q >> 1, k-step q & 1) of the K-tile's BN / 16 blocks
            wsrc[i] = reinterpret_cast<const char*>(p.W) + (((size_t)(n0 / 16 + (q >> 1)) * (p.K / 64)) * 1024 + lane * 16 + (q & 1) * 8) * 2;
        } else {
            const int slot = i * NTHR + tid, row = slot >> 3, kc = (slot & 7) ^ ((row >> 1) & 7);
            wsrc[i] = reinterpret_cast<const char*>(p.W) + ((size_t)(n0 + row) * p.ldw + kc * CE) * ESZ;
        }
    }
    // SPREAD: issue the next tile's LDS-DMA pieces between the MFMA groups (pays on the 8-wave 256^2 tile: +3..5 %;
    // on the 4-wave 128^2 tile with 2-3 workgroups per CU it measured -19 %, so that one issues them up front)
    constexpr bool SPREAD = (BM == 256);
    auto a_stage = [&](int i) -> char* { return smem + i * STAGE; };              // [A0 W0 | A1 W1]
    auto w_stage = [&](int i) -> char* { return smem + i * STAGE + A_BYTES; };
    auto issue_a = [&](int kt) {
        char* sa = a_stage(NSTAGE > 2 ? kt % NSTAGE : kt & 1);
        const int ko = kt * 128;                       // bytes: one K-tile = 128-byte rows
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sa + (i * NTHR + wid * 64) * 16), 16, 0, 0);
    };
    // one 1-KiB-per-wave piece of tile kt (pieces 0..A_IT-1 = A, A_IT.. = W); issued BETWEEN MFMA groups so the
    // ~100-cycle issue cost of each LDS-DMA hides under the matrix pipe instead of delaying the first MFMA of the tile
    auto issue_piece = [&](int kt, int piece) {
        const int ko = kt * 128;                       // bytes: one K-tile = 128-byte rows
        if (piece < A_IT) {
            char* sa = a_stage(kt & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[piece] + ko),
                                             (__attribute__((address_space(3))) void*)(sa + (piece * NTHR + wid * 64) * 16), 16, 0, 0);
        } else {
            const int i = piece - A_IT;
            char* sw = w_stage(kt & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sw + (i * NTHR + wid * 64) * 16), 16, 0, 0);
        }
    };
    const bool w_nt = WT && (p.w_tiled & 2);           // A/B: non-temporal hint on the weight DMA (a stream ONE workgroup reads once)
    auto issue_w = [&](int kt) {
        char* sw = w_stage(NSTAGE > 2 ? kt % NSTAGE : kt & 1);
        const int ko = kt * (WT ? 2048 : 128);         // bytes: one K-tile = 128-byte rows / one 2 KB block per 16-row tile
        if (w_nt) {
#pragma unroll
            for (int i = 0; i < W_IT; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + ko),
                                                 (__attribute__((address_space(3))) void*)(sw + (i * NTHR + wid * 64) * 16), 16, 0, 2);
            return;
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sw + (i * NTHR + wid * 64) * 16), 16, 0, 0);
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int aoff[TM], woff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = swz(wm * (BM / WM) + i * 16 + r, WT ? 2 * g : g);
#pragma unroll
    for (int j = 0; j < TN; ++j) woff[j] = WT ? ((wn * (BN / WN) / 16 + j) * 2) * 1024 + lane * 16 : swz(wn * (BN / WN) + j * 16 + r, g);
    constexpr int A_KS = WT ? 4 : 6, W_KS = WT ? 10 : 6;      // k-step 1: chunk 2g + 1 (byte 16) / the tile's second 1 KB image; else chunk g + 4 (byte 64)

    const int nk_all = p.K * ESZ / 128;
    const int kt0 = EPI == EPI_PARTIAL ? kchunk * (nk_all / p.ks) : 0;
    const int nk = EPI == EPI_PARTIAL ? kt0 + nk_all / p.ks : nk_all;          // this workgroup's K-tiles [kt0, nk)
    if constexpr (NSTAGE > 2) {
#pragma unroll
        for (int d = 0; d < NSTAGE - 1; ++d)
            if (kt0 + d < nk) { issue_a(kt0 + d); issue_w(kt0 + d); }
    } else { issue_a(kt0); issue_w(kt0); }
    for (int kt = kt0; kt < nk; ++kt) {
        if constexpr (NSTAGE > 2) {
            // tile kt landed: this thread's pieces by its own counted wait (the later tiles' A_IT + W_IT pieces each may stay in flight), everybody's
            // by the barrier — which also says every wave is done reading tile kt - 1, whose stage the DMA issued next overwrites
            constexpr int PPT = A_IT + W_IT;
            static_assert((NSTAGE - 2) * PPT < 64, "vmcnt is a 6-bit counter");
            const int ahead = min(nk - 1 - kt, NSTAGE - 2);
            if (NSTAGE >= 5 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 5 ? 3 * PPT : 0) : "memory");
            else if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NSTAGE - 1 < nk) { issue_a(kt + NSTAGE - 1); issue_w(kt + NSTAGE - 1); }
        } else {
            __syncthreads();                               // tile kt landed (vmcnt(0) + barrier); everyone is done with tile kt-1
        }
        if (tr && kt == 0 && threadIdx.x == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
        if (NSTAGE == 2 && !SPREAD && kt + 1 < nk) { issue_a(kt + 1); issue_w(kt + 1); }
        const bool more = SPREAD && kt + 1 < nk;
        const char* sa = a_stage(NSTAGE > 2 ? kt % NSTAGE : kt & 1);
        const char* sw = w_stage(NSTAGE > 2 ? kt % NSTAGE : kt & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // fragment reads run one pair of m-tiles AHEAD of the MFMAs that consume them (register double buffer),
            // pinned with sched_group_barrier so the LDS latency of pair p+1 hides under the 2*TN MFMAs of pair p
            bf16x8_t wf[TN], ac[2], an[2];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sw + (woff[j] ^ (ks << W_KS)));
            ac[0] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[0] ^ (ks << A_KS)));
            ac[1] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[1] ^ (ks << A_KS)));
            __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#pragma unroll
            for (int ip = 0; ip < TM / 2; ++ip) {
                if (ip + 1 < TM / 2) {
                    an[0] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[2 * ip + 2] ^ (ks << A_KS)));
                    an[1] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[2 * ip + 3] ^ (ks << A_KS)));
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[2 * ip + ii][j] = mma_step<FP8>(wf[j], ac[ii], acc[2 * ip + ii][j]);
                __builtin_amdgcn_sched_group_barrier(0x008, (FP8 ? 4 : 2) * TN, 0);
                if (more) {
                    constexpr int PPG = (A_IT + W_IT) / TM;            // pieces per MFMA group
#pragma unroll
                    for (int q = 0; q < PPG; ++q) issue_piece(kt + 1, (ks * (TM / 2) + ip) * PPG + q);
                    __builtin_amdgcn_sched_group_barrier(0x020, PPG, 0);
                }
                if (ip + 1 < TM / 2) { ac[0] = an[0]; ac[1] = an[1]; }
            }
        }
    }
    if constexpr (EPI == EPI_PARTIAL) {
        // fp32 accumulators straight to the chunk's partial rows: lane (r, g) holds 4 consecutive columns of row wm*.. + i*16 + r
        float* pr = p.part + (size_t)kchunk * SK_ROWS * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (BM / WM) + i * 16 + r;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float* dst = pr + (size_t)m * p.N + n0 + wn * (BN / WN) + j * 16 + g * 4;
                    // A/B (w_tiled bits 3 / 4): the partial rows leave as non-temporal / write-through (sc1) stores — the next kernel reads them, and
                    // dirty lines left in the L2s are written back at the kernel boundary (MI355X_MICROARCH.md "boundary": + bytes / 6 TB/s)
                    if (p.w_tiled & 8) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4_t*>(dst));
                    else if (p.w_tiled & 16) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(acc[i][j]) : "memory");
                    else *reinterpret_cast<f32x4_t*>(dst) = acc[i][j];
                }
            }
        }
        return;
    }
    constexpr int OUTW = GLU ? BN / 2 : BN;
    constexpr int CPR = OUTW / 8;
    const int on0 = GLU ? n0 / 2 : n0;
    constexpr int OIT = BM * CPR / NTHR;
    static_assert(BM * CPR % NTHR == 0, "output pieces must divide over the workgroup");
    // residual pieces requested up front, all in flight under the epilogue math — one dependent HBM round trip per piece
    // in the store loop made it 12 us on the residual GEMMs (now 2.9)
    uint4 rres[EPI == EPI_RESIDUAL ? OIT : 1];
    if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int it = 0; it < OIT; ++it) {
            const int c = it * NTHR + tid, row = c / CPR, ch = c - row * CPR;
            const int m = min(m0 + row, p.M - 1);
            rres[it] = *reinterpret_cast<const uint4*>(p.R + (size_t)m * p.ldr + on0 + ch * 8);
        }
    }
    // the lane's 4 bias values per column tile, fetched ONCE (8 bytes per tile) before the barrier: inside the loops
    // below the compiler cannot hoist them past the LDS stores, and 128 dependent 2-byte loads per lane were most of a
    // 10 us epilogue (tools/gemm_trace.py)
    float bv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        uint2 b2 = make_uint2(0u, 0u);
        if (!FP8 && !GLU && p.bias) b2 = *reinterpret_cast<const uint2*>(p.bias + n0 + wn * (BN / WN) + j * 16 + g * 4);   // (the fp8 path is the bias-free LLM)
        bv[j][0] = bflo(b2.x); bv[j][1] = bfhi(b2.x); bv[j][2] = bflo(b2.y); bv[j][3] = bfhi(b2.y);
    }
    // fp8: scale of the lane's TM rows (A rows) and of its 4 columns per column tile (W rows)
    float sar[FP8 ? TM : 1], swc[FP8 ? TN : 1][4];
    if constexpr (FP8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) sar[i] = p.sa[min(m0 + wm * (BM / WM) + i * 16 + r, p.M - 1)];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(p.sw + n0 + wn * (BN / WN) + j * 16 + g * 4);
            swc[j][0] = t4[0]; swc[j][1] = t4[1]; swc[j][2] = t4[2]; swc[j][3] = t4[3];
        }
    }
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[2] = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue: fp32 bias/activation -> bf16 -> LDS rows -> 16-byte stores ----
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
                    if constexpr (FP8) x *= sar[i] * swc[j][q];
                    else x += bv[j][q];
                    if (EPI == EPI_QUICKGELU) x = x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));   // x*sigmoid(1.702x)
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
                    if constexpr (FP8) { gt *= sar[i] * swc[2 * jj][q]; up *= sar[i] * swc[2 * jj + 1][q]; }
                    v[q] = gt * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;   // silu(g)*u
                }
                *reinterpret_cast<uint2*>(smem + mrow * OSTRIDE + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
    }
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int it = 0; it < OIT; ++it) {
        const int c = it * NTHR + tid;
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
        // non-temporal: the tile is consumed by the NEXT kernel, not by this one (+2-3 % on the K = 1024 shapes)
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + on0 + ch * 8));
    }
    if (tr && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[4] = __builtin_amdgcn_s_memrealtime(); }
}


template <int BM, int BN, int WM, int WN, bool FP8>
int launch_glds(const GemmArgs& p, int epi, hipStream_t s) {
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int LOOPB = 2 * STAGE;
    constexpr int OBYTES = BM * (BN * 2 + 16);
    constexpr size_t lds = (LOOPB > OBYTES) ? LOOPB : OBYTES;
    constexpr int NTHR = WM * WN * 64;
    const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
    static LdsGrant g0, g1, g2, g3;
    if (!grant_dynamic_lds(g0, reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_NONE, FP8>), (int)lds) ||
        !grant_dynamic_lds(g1, reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_RESIDUAL, FP8>), (int)lds) ||
        !grant_dynamic_lds(g2, reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_QUICKGELU, FP8>), (int)lds) ||
        !grant_dynamic_lds(g3, reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_SWIGLU, FP8>), (int)lds)) return TRACE_ERR_HIP;
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_NONE, FP8>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        case EPI_RESIDUAL: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_RESIDUAL, FP8>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        case EPI_QUICKGELU: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_QUICKGELU, FP8>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_SWIGLU, FP8>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        default: return TRACE_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

}  // namespace

int g_gemm_variant = 0;   // 0 = auto, 2 = 128^2 tiles, 3 = this file's 256^2 kernel, 4 = the loader-wave 256^2 kernel, 5 / 6 = the persistent one
                          // (gemm_pers.hip; 6 = static tile deal) — tests / microbench
extern std::atomic<int> g_gemm_pers_static;   // (written on the launch path by both pipeline host threads, always with the value it holds: atomic so that this is defined behaviour)
int g_gemm_w4 = -1;          // non-residual 256^2 shapes on gemm_w4.hip in auto mode: 1 / 0 (trace_op_set_gemm_variant(530 + x)); -1 = TRACE_GEMM_W4 from the environment, else on
static bool gemm_w4_enabled() {
    static const int env = getenv("TRACE_GEMM_W4") ? (atoi(getenv("TRACE_GEMM_W4")) != 0) : 1;      // (read once; the pipeline's two host threads both come through here)
    return g_gemm_w4 < 0 ? env != 0 : g_gemm_w4 != 0;
}
int g_gemm_resid_pers = 0;   // 1: residual shapes also run on the persistent kernel in auto mode (trace_op_set_gemm_variant(520 + x); A/B runs)

// split-K partial-row GEMM for decode batches above SKINNY_ROWS: M <= 128 rows (one row panel), 128 x BN tiles, ks chunks of K
template <int EPI, bool WT, int NSTAGE, int BN = 128, int WM = 2, int WN = 2>
static int launch_dec(const GemmArgs& p, int nblk, hipStream_t s) {
    constexpr int STAGE = (128 + BN) * 128, LOOPB = NSTAGE * STAGE, OBYTES = 128 * (BN * 2 + 16), LDSB = LOOPB > OBYTES ? LOOPB : OBYTES;
    static LdsGrant grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(gemm_glds_kernel<128, BN, WM, WN, EPI, false, WT, NSTAGE>), LDSB)) return TRACE_ERR_HIP;
    hipLaunchKernelGGL((gemm_glds_kernel<128, BN, WM, WN, EPI, false, WT, NSTAGE>), dim3(nblk), dim3(256), LDSB, s, p);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
// Tile shape of the partial-row GEMM (A/B: trace_op_set_gemm_variant(740 + x)): 0 = 128 x 128 on 2 x 2 waves, 4-stage ring (round 3); 1 / 2 / 3 = 128 x 64 on
// 4 x 1 waves with a 5- / 3- / 4-stage ring (24 KB stages: half the partial-row bytes per product at the same number of workgroups, K loops twice as long;
// the 3-stage form fits two workgroups per CU)
int g_partial_cfg = 0;
int g_partial_wgs = 0;      // workgroup target of gemm_partial_ks (0 = TRACE_PARTIAL_WGS from the environment, else 192; A/B: trace_op_set_gemm_variant(800 + n / 32))
static int partial_bn() { return g_partial_cfg ? 64 : 128; }
static int launch_partial(const GemmArgs& p, hipStream_t s) {
    const int bn = (p.w_tiled & 1) ? partial_bn() : 128;
    if (p.M < 1 || p.M > SK_ROWS || p.N % bn || p.K % BK || p.fp8 || !p.part || p.ks < 1 || (p.K / BK) % p.ks) return TRACE_ERR_ARG;
    if ((p.lda % 8) || (p.ldw % 8)) return TRACE_ERR_ARG;
    const int nblk = ((p.M + 127) / 128) * (p.N / bn) * p.ks;      // above 128 rows: two row panels (neighbours in the tile order: the second reads its weight tile from L2)
    if (!(p.w_tiled & 1)) return launch_dec<EPI_PARTIAL, false, 2>(p, nblk, s);
    if (bn == 64) return g_partial_cfg == 1 ? launch_dec<EPI_PARTIAL, true, 5, 64, 4, 1>(p, nblk, s) : g_partial_cfg == 2 ? launch_dec<EPI_PARTIAL, true, 3, 64, 4, 1>(p, nblk, s)
                                                                                                  : launch_dec<EPI_PARTIAL, true, 4, 64, 4, 1>(p, nblk, s);
    return (p.w_tiled & 4) ? launch_dec<EPI_PARTIAL, true, 4>(p, nblk, s) : launch_dec<EPI_PARTIAL, true, 2>(p, nblk, s);
}
// gate|up of a wide decode step: [M <= 128, K] x tiled W -> SwiGLU -> bf16, one row panel of 128x128 tiles
static int launch_swiglu_tiled(const GemmArgs& p, hipStream_t s) {
    if (p.M < 1 || p.M > SK_ROWS || p.N % 128 || p.K % BK || p.fp8 || (p.lda % 8) || (p.ldc % 8)) return TRACE_ERR_ARG;
    const int nblk = ((p.M + 127) / 128) * (p.N / 128);
    return (p.w_tiled & 4) ? launch_dec<EPI_SWIGLU, true, 4>(p, nblk, s) : launch_dec<EPI_SWIGLU, true, 2>(p, nblk, s);
}
// K-chunks for the partial-row GEMM (sized for one row panel): enough workgroups ((N / tile width) x ks) to reach the target, chunks of whole K-tiles, at least 4
// K-tiles per chunk (a shorter K loop is all prologue).  Target 192 since round 6 (was 256): the 7B qkv product (48 column tiles) is then cut in 4 chunks = 192 workgroups
// in one round instead of 8 = 384 in one and a half, with half the partial-row bytes; o / down (32 tiles) keep 8 chunks = 256.  Wide step, ms per 128-sequence step at
// ctx 1968 (profiles/r06_decode_splitk_ab.txt): target 256 11.01, 192 10.59, 160 10.41 (= 192), 128 (o / down at 4 chunks) 10.72, 96 10.96, 64 12.27.
// TRACE_PARTIAL_WGS overrides the workgroup target (tuning runs).
int gemm_partial_ks(int N, int K) {
    static const int env_target = getenv("TRACE_PARTIAL_WGS") ? atoi(getenv("TRACE_PARTIAL_WGS")) : 192;
    const int target = g_partial_wgs > 0 ? g_partial_wgs : env_target;
    const int tiles = N / partial_bn(), nk = K / BK;
    int ks = 1;
    while (tiles * ks < target && nk % (ks * 2) == 0 && nk / (ks * 2) >= 4) ks *= 2;
    return ks;
}

int launch_gemm_bf16(const GemmArgs& p, int epi, hipStream_t s) {
    if (epi == EPI_PARTIAL) return launch_partial(p, s);
    if (p.w_tiled) return epi == EPI_SWIGLU ? launch_swiglu_tiled(p, s) : TRACE_ERR_ARG;
    if (p.M <= 0 || p.N % BN || p.K % BK || p.K < BK) return TRACE_ERR_ARG;
    if (p.fp8 && (p.K % 128 || (p.lda % 16) || (p.ldw % 16) || !p.sa || !p.sw || p.bias || epi == EPI_QUICKGELU)) return TRACE_ERR_ARG;
    if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 8)) return TRACE_ERR_ARG;
    if (epi == EPI_RESIDUAL && (!p.R || (p.ldr % 8))) return TRACE_ERR_ARG;
    {
        int v = g_gemm_variant;
        if (v == 0) {
            // 256^2 tiles (one workgroup per CU, loader-wave kernel) when they fill at least 70 % of their rounds of the 256 CUs;
            // otherwise the 128^2 kernel (two workgroups per CU).  Measured (us, 128^2 vs loader-wave 256^2; tools/gemm_variant_check.py):
            // 192 tiles = 0.75 round (single-prompt qkv) 101 vs 84; 384 = 1.5 rounds (paired qkv) 182 vs 176; 128 = half a round
            // (single-prompt o / down) 64 vs 79 and 211 vs 250.  All three kernels give the same bits (same K order).
            const long blocks256 = (long)((p.M + 255) / 256) * (p.N / 256);
            const long rounds = (blocks256 + 255) / 256;
            v = (p.N % 256 == 0 && p.M >= 1024 && blocks256 * 10 >= rounds * 256 * 7) ? 3 : 2;
        }
        // 256^2 tiles run on the loader-wave kernels (same results bit for bit as this file's kernel, which variant 3 forces for A/B runs):
        // gemm_ldr.hip (+20-24 % on the K = 1024 ViT shapes) and, where there is no residual to fetch, its persistent form gemm_pers.hip
        // (tools/gemm_pers_ab.py, interleaved medians vs gemm_ldr over several boxes, both with their L2 touches: fc1 + QuickGELU -8 .. -11 %,
        // ViT qkv -4 .. -6 %, prefill gate|up -3 .. -7 %, prefill qkv 0 .. -5 %).  With a residual the one-workgroup-per-tile kernel stays: its
        // LDS-staged epilogue reads the touched residual lines as full rows; the persistent kernel's two-pass register epilogue measured
        // -5 .. +14 % against it depending on the box.
        const bool pers_ok = p.N % 256 == 0 && !p.fp8 && p.K >= 128 && (long)p.M * p.ldc < (1L << 30) &&
                             (epi != EPI_RESIDUAL || (long)p.M * p.ldr < (1L << 30));
        if (g_gemm_variant == 8 && pers_ok && p.K >= 192 && (long)p.M * p.lda < (1L << 31) && (long)p.N * p.ldw < (1L << 31)) {
            g_gemm_pers_static = 0;
            return launch_gemm_w4(p, epi, s);                  // the 4-wave persistent kernel (A/B runs)
        }
        if ((g_gemm_variant >= 5 && g_gemm_variant <= 7) && pers_ok) {
            g_gemm_pers_static = g_gemm_variant - 5;        // 5 ticketed, 6 static deal, 7 one workgroup per tile
            return launch_gemm_pers(p, epi, s);
        }
        if (p.N % 256 == 0 && (g_gemm_variant == 4 || (g_gemm_variant == 0 && v == 3))) {
            if (g_gemm_variant == 0 && pers_ok && (epi != EPI_RESIDUAL || g_gemm_resid_pers)) {
                g_gemm_pers_static = 0;
                // without a residual: the 4-wave form of the persistent kernel (gemm_w4.hip; same bits), unless switched off
                // (trace_op_set_gemm_variant(530) / TRACE_GEMM_W4=0: A/B runs)
                const bool w4 = gemm_w4_enabled() && epi != EPI_RESIDUAL && p.K >= 192 && (long)p.M * p.lda < (1L << 31) && (long)p.N * p.ldw < (1L << 31);
                const int rc = w4 ? launch_gemm_w4(p, epi, s) : launch_gemm_pers(p, epi, s);
                // no ticket counters for this stream and none can be made inside a capture: the one-workgroup-per-tile kernel gives the same bits
                if (rc != TRACE_ERR_STATE) return rc;
            }
            return launch_gemm_ldr(p, epi, s);
        }
        if (v == 3 && p.N % 256 == 0) return p.fp8 ? launch_glds<256, 256, 2, 4, true>(p, epi, s) : launch_glds<256, 256, 2, 4, false>(p, epi, s);
    }
    return p.fp8 ? launch_glds<128, 128, 2, 2, true>(p, epi, s) : launch_glds<128, 128, 2, 2, false>(p, epi, s);
}
