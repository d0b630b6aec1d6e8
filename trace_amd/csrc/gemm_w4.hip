// The 256x256 persistent GEMM with FOUR waves per workgroup, each holding a 128x128 sub-tile (round 5).
//
// Why: gemm_pers.hip's 8 MFMA waves hold 128x64 each, so a K-tile costs 192 KB of fragment reads (12 KB per wave and k-step) for its 2048 MFMA
// cycles per SIMD; a 128x128 sub-tile needs 16 KB per wave and k-step for TWICE the MFMAs — 128 KB per K-tile, a third less LDS traffic per flop
// (tools/gemm_lib_yardstick.py: the vendor library's kernels for these shapes are 256x256x64 tiles on 4 waves and run 2-9 % ahead of gemm_pers).
// The price: 256 accumulator registers per lane (they live in AGPRs that only the inline asm names: gemm_agpr.h), ONE wave per SIMD — nothing
// hides a stall, so the wave's own instruction stream is the schedule — and no spare waves to do the loading: the four waves issue the LDS-DMA
// pieces themselves, between their MFMAs.
//
// K loop (per wave, K-tile q in stage q & 1):
//   k-step 0: 64 MFMAs on fragment buffer 0; under them the 16 fragment reads of k-step 1 (same stage) into buffer 1;
//   s_waitcnt lgkmcnt(0) — every fragment of the stage is now in registers; s_waitcnt vmcnt(..) — this wave's pieces of K-tile q + 1 have
//   landed; s_barrier — so have everyone's, and the stage of K-tile q is free;
//   k-step 1: 64 MFMAs on buffer 1; under them this wave's 16 LDS-DMA pieces of K-tile q + 2 into the freed stage, and the 16 fragment reads
//   of K-tile q + 1's k-step 0 into buffer 0.
// One hand-over barrier per K-tile (plus one at its start that only re-aligns the four waves), half to one K-tile period of load slack, the stream of K-tiles runs
// across output tiles.  Waits are counted by hand (s_waitcnt vmcnt(N): 0 in a tile, the epilogue's store count behind one): every VMEM operation of the kernel has a
// fixed place in the wave's instruction stream, and tests/test_kernel_resources.py fails a build that spills (a scratch access is a VMEM operation the counts do not know).
// All LDS reads are inline asm: hipcc's wait-count pass cannot tell a ds_read from the LDS-DMA writes in flight and would put s_waitcnt vmcnt(0)
// in front of every one (cdna_hip_programming.md, "Pipelining across barriers"; attn.hip's 192-row kernel met the same thing).
// Same LDS image (XOR swizzle on the DMA source), same K order, same fp32 -> bf16 roundings as gemm_pers.hip / gemm_ldr.hip: bit-identical results.
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "gemm_tilewalk.h"
#include "gemm_agpr.h"

namespace {

using tilewalk::BM; using tilewalk::BN; using tilewalk::BK; using tilewalk::CTR_STRIDE;
using tilewalk::tile_coords; using tilewalk::Sched; using tilewalk::make_sched; using tilewalk::swap16;
constexpr int NTHR = 256;
constexpr int TM = 8, TN = 8;                     // 16x16 fragments of a wave's 128x128 sub-tile
constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
constexpr int CTL_OFF = 2 * STAGE;                // int s_next[2] | 2 x 512-byte bias rows
constexpr int BIAS_OFF = CTL_OFF + 64;
constexpr int TOUCH_OFF = BIAS_OFF + 2 * 512;     // 256 bytes: where the L2 touches land (never read)
constexpr int LDS_BYTES = TOUCH_OFF + 256;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ uint32_t swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ uint32_t lds_u32(char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)p; }

template <int OFF>
__device__ __forceinline__ void lds_rd(bf16x8_t& d, uint32_t a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(OFF));
}
// s_waitcnt lgkmcnt(0) carrying the 16 fragments as tied operands: no use (or copy) of one can be scheduled before the wait
__device__ __forceinline__ void lgkm0(bf16x8_t (&a)[TM], bf16x8_t (&w)[TN]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
}
__device__ __forceinline__ void mfma_acc(int n, const bf16x8_t& w, const bf16x8_t& a) {      // n is a constant after unrolling
    switch (n) {
#define X(N, R0, R1, R2, R3) case N: mfma_acc_##N(w, a); break;
        W4_ACC_LIST(X)
#undef X
    }
}
__device__ __forceinline__ void mfma_new(int n, const bf16x8_t& w, const bf16x8_t& a) {
    switch (n) {
#define X(N, R0, R1, R2, R3) case N: mfma_new_##N(w, a); break;
        W4_ACC_LIST(X)
#undef X
    }
}
__device__ __forceinline__ f32x4_t acc_get(int n) {
    switch (n) {
#define X(N, R0, R1, R2, R3) case N: return acc_get_##N();
        W4_ACC_LIST(X)
#undef X
    }
    return f32x4_t{0.f, 0.f, 0.f, 0.f};
}
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)

// Four packed fragments of one 16-row m-tile (64 consecutive output columns; pk[f] = lane's 4 columns of fragment f) -> two 16-byte stores, each
// 8 rows x one full 128-byte line (gemm_pers.hip epilogue_rows: v_permlane16_swap makes 8 consecutive columns per lane, DPP row_ror:8 trades a
// piece between rows r and r + 8).  `off` = byte offset of the lane's piece in the first store; the second is 8 rows further.
__device__ __forceinline__ void line_pieces(uint32_t (&pk)[4][2], u32x4& A, u32x4& B, bool hi8) {
    u32x4 P[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        swap16(pk[2 * h][0], pk[2 * h + 1][0]);
        swap16(pk[2 * h][1], pk[2 * h + 1][1]);
        P[h] = u32x4{pk[2 * h][0], pk[2 * h][1], pk[2 * h + 1][0], pk[2 * h + 1][1]};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t snd = hi8 ? P[0][e] : P[1][e];
        const uint32_t rcv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)snd, 0x128, 0xf, 0xf, false);     // row_ror:8 = lane r ^ 8
        A[e] = hi8 ? rcv : P[0][e];
        B[e] = hi8 ? P[1][e] : rcv;
    }
}

template <int EPI, int OPT>
__global__ __launch_bounds__(NTHR) void gemm_w4_kernel(GemmArgs p, int* ctr, int dynamic) {
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Sched sc = make_sched(p);
    const int ntn = sc.ntn, ntm = sc.ntm, nk = sc.nk, cnt = sc.cnt, base = sc.base, nwg = sc.nwg, xcd = sc.xcd;
    const int wm = wid >> 1, wn = wid & 1;
    const uint32_t lds0 = lds_u32(smem);
    const uint32_t y = swz(lane & 15, lane >> 4);
    const uint32_t fa0 = lds0 + wm * (128 * 128) + y, fw0 = lds0 + A_BYTES + wn * (128 * 128) + y;      // stage 0, k-step 0 (k-step 1: ^ 64; stage 1: + STAGE)
    const bool leader = tid == 0;
    const bool bias_wave = !GLU && wid == 3;

    bf16x8_t af[2][TM], wf[2][TN];
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.M * p.ldc * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(EPI == EPI_RESIDUAL ? p.R : p.C), 0,
                                                                         p.M * (EPI == EPI_RESIDUAL ? p.ldr : p.ldc) * 2, 0x00020000);

    // ---- the DMA cursor: the K-tile stream runs two K-tiles ahead of the MFMAs, across output tiles ----
    // this wave's 16 pieces of a K-tile: A rows wid*64 .. +63 and W rows wid*64 .. +63 of the tile, 8 rows x 128 bytes each
    // Pieces go through buffer descriptors (buffer_load_dwordx4 .. lds: scalar base + per-lane 32-bit row offset + SCALAR K offset — no address
    // arithmetic on the VALU per piece; rows of the last row panel beyond M are out of range and arrive as zeros, no clamp).
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)(uint32_t)((long)p.M * p.lda * 2), 0x00020000);      // (the launcher keeps both extents below 4 GB)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, (int)(uint32_t)((long)p.N * p.ldw * 2), 0x00020000);
    uint32_t voA[8], voW[8];                                 // per-lane byte offsets of the pieces' rows (swizzled 16-byte chunk included)
    constexpr bool SAFE = (OPT & 2) != 0;                    // check build: every counted s_waitcnt vmcnt(N) below becomes vmcnt(0).  Slower, and correct whatever the
                                                             // counts are worth: tests/test_gpu_kernels.py compares it bit for bit with the counted build, so a hipcc upgrade that moves a
                                                             // VMEM operation across one of the counts shows up as a difference instead of as a rare wrong tile
    constexpr bool TOUCH = (OPT & 4) != 0;                   // wave 0 touches the A lines of K-tile kt + 3 into L2 (gemm_pers.hip's L2 touches), one instruction per K-tile
    const bool toucher = TOUCH && wid == 0;
    uint32_t voT = 0;
    uint2 bias2 = make_uint2(0u, 0u);                        // bias_wave: the cursor tile's bias row, 4 columns per lane (zeros without a bias)
    int d_kt = 0;                                            // next K-tile of the cursor's tile to issue
    auto dma_setup = [&](int li_) __attribute__((always_inline)) {
        int tm_, tn_;
        tile_coords(base + li_, ntm, ntn, tm_, tn_);
        const int m0_ = tm_ * BM, n0_ = tn_ * BN;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = wid * 64 + j * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            voA[j] = (uint32_t)(m0_ + row) * (uint32_t)(p.lda * 2) + kc * 16;
            voW[j] = (uint32_t)(n0_ + row) * (uint32_t)(p.ldw * 2) + kc * 16;
        }
        if (TOUCH) voT = (uint32_t)(m0_ + (tn_ & 3) * 64 + lane) * (uint32_t)(p.lda * 2);
        if (bias_wave && p.bias) bias2 = *reinterpret_cast<const uint2*>(p.bias + n0_ + lane * 4);
        d_kt = 0;
        W4_FENCE();                                          // the counted waits below assume this load is OLDER than the pieces that follow
    };
    auto dma_piece = [&](int stage, int J) __attribute__((always_inline)) {                 // piece J of the cursor's K-tile d_kt into `stage` (J < 8: A, else W)
        char* dst = smem + stage * STAGE + (J < 8 ? 0 : A_BYTES) + (wid * 64 + (J & 7) * 8) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(J < 8 ? ars : wrs, (__attribute__((address_space(3))) void*)dst, 16, J < 8 ? voA[J & 7] : voW[J & 7], d_kt * 128, 0, 0);
    };
    auto dma_touch = [&]() __attribute__((always_inline)) {      // rides behind the pieces of K-tile d_kt: the in-order counter is waited down to it, not through it
        // ALWAYS one instruction per batch of pieces (the toucher's waits count it): past the tile's last K-tile it touches that one again
        if (toucher)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(smem + TOUCH_OFF), 4, voT, min(d_kt + 3, nk - 1) * 128, 0, 0);
    };
    // the cursor tile's bias row into LDS (row parity = the tile's), right behind a full vmcnt drain and in front of a barrier
    auto bias_publish = [&](int parity) __attribute__((always_inline)) {
        if (bias_wave) asm volatile("ds_write_b64 %0, %1" :: "v"(lds0 + BIAS_OFF + parity * 512 + lane * 8), "v"(bias2) : "memory");
    };

    // fragment reads of (stage, k-step) into buffer b: slot s = 0..7 an A fragment, 8..15 a W fragment
    auto frag_rd = [&](int b, int stage, int ks, int s) __attribute__((always_inline)) {
        const uint32_t a = (s < 8 ? fa0 : fw0) + stage * STAGE;
        const uint32_t ax = ks ? (a ^ 64u) : a;
        switch (s & 7) {
            case 0: lds_rd<0 * 2048>(s < 8 ? af[b][0] : wf[b][0], ax); break;
            case 1: lds_rd<1 * 2048>(s < 8 ? af[b][1] : wf[b][1], ax); break;
            case 2: lds_rd<2 * 2048>(s < 8 ? af[b][2] : wf[b][2], ax); break;
            case 3: lds_rd<3 * 2048>(s < 8 ? af[b][3] : wf[b][3], ax); break;
            case 4: lds_rd<4 * 2048>(s < 8 ? af[b][4] : wf[b][4], ax); break;
            case 5: lds_rd<5 * 2048>(s < 8 ? af[b][5] : wf[b][5], ax); break;
            case 6: lds_rd<6 * 2048>(s < 8 ? af[b][6] : wf[b][6], ax); break;
            default: lds_rd<7 * 2048>(s < 8 ? af[b][7] : wf[b][7], ax); break;
        }
    };

    int li = sc.slot, n = 0, q = 0;
    int li_next = li + nwg;                                  // static deal; the dynamic walk overwrites it from the ticket
    int ticket = 0;
    const uint32_t s_next_lds = lds0 + CTL_OFF;

    // ---- prologue: K-tiles 0 and 1 of the first tile; its bias row; first fragments ----
    dma_setup(li);
#pragma unroll
    for (int J = 0; J < 16; ++J) dma_piece(0, J);
    d_kt = 1;
#pragma unroll
    for (int J = 0; J < 16; ++J) dma_piece(1, J);
    if (TOUCH) dma_touch();
    d_kt = 2;
    bool d_next = false;                                     // the cursor has moved on to the next tile
    bool has_next = false;
    if (SAFE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (toucher) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // K-tile 0 (the bias load is older still)
    bias_publish(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < 16; ++s) frag_rd(0, 0, 0, s);
    lgkm0(af[0], wf[0]);

    int pend = 0;                                            // 1: an epilogue's stores are younger than the pieces the next barrier waits for

    // k-step 1 of K-tile q (buffer 1): ISSUE: this wave's pieces of K-tile q + 2 -> stage st; MORE: fragments of K-tile q + 1, k-step 0 -> buffer 0
    auto kstep1 = [&](auto issue_c, auto more_c, const int st) __attribute__((always_inline)) {
        constexpr bool ISSUE = decltype(issue_c)::value, MORE = decltype(more_c)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                mfma_acc(i * TN + j, wf[1][j], af[1][i]);
                const int s = i * TN + j;
                // Fragment reads in the first half; the 16 pieces one every 4 MFMAs over the WHOLE k-step: issued in a burst (all in the first 16 slots — the longest
                // landing time) the VMEM queue backs up and the wave's MFMAs wait behind it: 8192^3 795 us vs 753 spread (profiles/r05_gemm_w4_ab.txt, run 3)
                if (MORE && (s & 1) == 0 && s < 32) frag_rd(0, st ^ 1, 0, s >> 1);
                if (ISSUE && (s & 3) == 1) { dma_piece(st, s >> 2); W4_FENCE(); }
            }
        W4_FENCE();
        if (ISSUE) { if (TOUCH) dma_touch(); ++d_kt; }
        if (MORE) lgkm0(af[0], wf[0]);
    };
    auto ktile = [&](auto first_c, const int kt) __attribute__((always_inline)) {
        const int st = q & 1;
        // ---------------- k-step 0: buffer 0; reads of k-step 1 -> buffer 1 ----------------
        W4_FENCE();
        // A second barrier per K-tile: nothing is handed over here, the four waves just re-align.  Interleaved A/B on four boxes: +1-4 % on every shape
        // (profiles/r05_gemm_w4_knockouts.txt; free-running, one wave per SIMD, they drift apart and meet the hand-over barrier at different times).
        if (!(OPT & 1)) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (decltype(first_c)::value) mfma_new(i * TN + j, wf[0][j], af[0][i]);
                else mfma_acc(i * TN + j, wf[0][j], af[0][i]);
                const int s = i * TN + j;
                if ((s & 1) == 0 && s < 32) frag_rd(1, st, 1, s >> 1);
            }
        W4_FENCE();
        lgkm0(af[1], wf[1]);
        if (pend == 0) {
            // every piece issued so far has landed — and the ticket drawn a K-tile ago (tied operand: no use of it can be scheduled before this wait)
            if (toucher && !SAFE) asm volatile("s_waitcnt vmcnt(1)" : "+v"(ticket) :: "memory");      // (all but the touch behind the pieces; without one: a piece more than needed... never fewer)
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) :: "memory");
            if (kt == nk - 1 && d_next) bias_publish((n + 1) & 1);      // (the set-up that loaded it was a K-tile ago; this is the full drain behind it)
            if (kt == 1 && leader) {                         // the tile after this one, for everyone to read behind this barrier
                const int v = dynamic ? nwg + ticket : li + nwg;
                asm volatile("ds_write_b32 %0, %1" :: "v"(s_next_lds + ((n + 1) & 1) * 4), "v"(v) : "memory");
                // the launch's last ticket on this XCD re-arms the counter (gemm_pers.hip); asm: hipcc's atomic optimizer would wait for the result it drops
                if (dynamic && ticket == cnt - 1) asm volatile("global_atomic_swap %0, %1, off" :: "v"(ctr + xcd * CTR_STRIDE), "v"(0) : "memory");
            }
        } else if (SAFE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (EPI == EPI_SWIGLU) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // (a toucher's touch is OLDER than the stores: these counts hold for it too,
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");                                //  they just wait for its touch as well)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pend = 0;
        __builtin_amdgcn_s_barrier();                        // K-tile q + 1 has landed; the stage of K-tile q is free
        W4_FENCE();
        // the ticket of this tile's successor: drawn here, read a K-tile later (older than the pieces issued below: never waited for on its own).  Inline asm: through
        // atomicAdd() hipcc's atomic optimizer folds the wave's lanes into one add and reads the result back at once — s_waitcnt vmcnt(0) right here
        if (kt == 0 && dynamic && leader)
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(ctr + xcd * CTR_STRIDE), "v"(1) : "memory");
        if (kt == 1) {
            int v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(s_next_lds + ((n + 1) & 1) * 4) : "memory");
            li_next = __builtin_amdgcn_readfirstlane(v);
            has_next = li_next < cnt;
        }
        // the cursor: K-tile q + 2 of the stream (nk >= 3: it is in this tile or the next one)
        bool issue = true;
        if (d_kt >= nk) {
            if (!d_next && has_next) { dma_setup(li_next); d_next = true; }
            else issue = false;
        }
        const bool more = kt + 1 < nk || has_next;           // there is a K-tile q + 1 to read fragments of
        if (issue) kstep1(std::true_type{}, std::true_type{}, st);
        else if (more) kstep1(std::false_type{}, std::true_type{}, st);
        else kstep1(std::false_type{}, std::false_type{}, st);
    };

    while (true) {
        int tm, tn;
        tile_coords(base + li, ntm, ntn, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        ktile(std::true_type{}, 0);
        ++q;
        for (int kt = 1; kt < nk; ++kt, ++q) ktile(std::false_type{}, kt);
        // ---------------- epilogue ----------------
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        W4_FENCE();
        const int g_e = lane >> 4;
        const int pcol = ((g_e & 1) << 4) | ((g_e >> 1) << 3);      // first of the lane's 8 consecutive columns within a 32-column group
        const bool hi8 = (lane & 8) != 0;
        const int mbase = m0 + wm * 128 + (lane & 15);
        if constexpr (GLU) {
            const int cbase = n0 / 2 + wn * 64 + pcol;
            const int coff = (mbase * p.ldc + cbase) * 2 + (hi8 ? 64 - 16 * p.ldc : 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint32_t pk[4][2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float v[4];
                    const f32x4_t gtv = acc_get(i * TN + 2 * jj), upv = acc_get(i * TN + 2 * jj + 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gt = gtv[e], up = upv[e];
                        v[e] = gt * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;
                    }
                    pk[jj][0] = pack2bf(v[0], v[1]);
                    pk[jj][1] = pack2bf(v[2], v[3]);
                }
                u32x4 A, B;
                line_pieces(pk, A, B, hi8);
                __builtin_amdgcn_raw_buffer_store_b128(A, crs, coff + i * 32 * p.ldc, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(B, crs, coff + i * 32 * p.ldc + 16 * p.ldc, 0, 2);
                W4_FENCE();
            }
        } else {
            const int cbase = n0 + wn * 128 + pcol;
            const int coff = (mbase * p.ldc + cbase) * 2 + (hi8 ? 64 - 16 * p.ldc : 0);
            const int roff = (mbase * p.ldr + cbase) * 2 + (hi8 ? 64 - 16 * p.ldr : 0);
            uint2 bp[TN];                                    // this lane's 8 x 4 bias values, packed
            {
                const uint32_t ba = lds0 + BIAS_OFF + (n & 1) * 512 + (wn * 128 + g_e * 4) * 2;
                asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:32\n\tds_read_b64 %2, %8 offset:64\n\tds_read_b64 %3, %8 offset:96\n\t"
                             "ds_read_b64 %4, %8 offset:128\n\tds_read_b64 %5, %8 offset:160\n\tds_read_b64 %6, %8 offset:192\n\tds_read_b64 %7, %8 offset:224\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(bp[0]), "=&v"(bp[1]), "=&v"(bp[2]), "=&v"(bp[3]), "=&v"(bp[4]), "=&v"(bp[5]), "=&v"(bp[6]), "=&v"(bp[7]) : "v"(ba) : "memory");
            }
            // quarter by quarter (4 m-tiles x 64 columns): with a residual, the quarter's 8 pieces of R are requested before its arithmetic
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int i0 = 0; i0 < TM; i0 += 4) {
                    u32x4 rr[4][2];
                    if (EPI == EPI_RESIDUAL) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            rr[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + h2 * 128 + (i0 + i) * 32 * p.ldr, 0, 0);
                            rr[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + h2 * 128 + (i0 + i) * 32 * p.ldr + 16 * p.ldr, 0, 0);
                        }
                        W4_FENCE();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint32_t pk[4][2];
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint2 b = bp[4 * h2 + j];
                            const f32x2_t b01 = {bflo(b.x), bfhi(b.x)}, b23 = {bflo(b.y), bfhi(b.y)};
                            const f32x4_t av = acc_get((i0 + i) * TN + 4 * h2 + j);
                            f32x2_t x01 = f32x2_t{av[0], av[1]} + b01;
                            f32x2_t x23 = f32x2_t{av[2], av[3]} + b23;
                            if (EPI == EPI_QUICKGELU) {
                                const f32x2_t t01 = x01 * -2.4554669595930157f, t23 = x23 * -2.4554669595930157f;
                                const f32x2_t d01 = f32x2_t{__builtin_amdgcn_exp2f(t01[0]), __builtin_amdgcn_exp2f(t01[1])} + 1.f;
                                const f32x2_t d23 = f32x2_t{__builtin_amdgcn_exp2f(t23[0]), __builtin_amdgcn_exp2f(t23[1])} + 1.f;
                                x01 = x01 * f32x2_t{__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1])};
                                x23 = x23 * f32x2_t{__builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
                            }
                            pk[j][0] = pack2bf(x01[0], x01[1]);
                            pk[j][1] = pack2bf(x23[0], x23[1]);
                        }
                        u32x4 o[2];
                        line_pieces(pk, o[0], o[1], hi8);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            u32x4 v = o[h];
                            if (EPI == EPI_RESIDUAL) {
                                const u32x4 r = rr[i][h];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = pack2bf(bflo(v[e]) + bflo(r[e]), bfhi(v[e]) + bfhi(r[e]));
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(v, crs, coff + h2 * 128 + (i0 + i) * 32 * p.ldc + h * 16 * p.ldc, 0, 2);
                        }
                        W4_FENCE();
                    }
                }
        }
        W4_FENCE();
        if (!has_next) break;
        pend = 1;
        li = li_next;
        d_next = false;
        ++n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, int OPT>
bool launch_opt(const GemmArgs& p, int nblk, int dynamic, int* ctr, hipStream_t s) {
    static LdsGrant grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(gemm_w4_kernel<EPI, OPT>), LDS_BYTES)) return false;      // refused (or a device id the grant table does not hold)
    hipLaunchKernelGGL((gemm_w4_kernel<EPI, OPT>), dim3(nblk), dim3(NTHR), LDS_BYTES, s, p, ctr, dynamic);
    return true;
}
template <int EPI>
bool launch_one(const GemmArgs& p, int nblk, int dynamic, int* ctr, hipStream_t s) {
    switch (p.opt & 7) {          // A/B builds (trace_op_set_gemm_variant(540 + opt)): bit 0 = without the re-aligning barrier, bit 2 = with the L2 touches of the A panel; 2 = the check build (vmcnt(0) waits)
        case 1: return launch_opt<EPI, 1>(p, nblk, dynamic, ctr, s);
        case 2: return launch_opt<EPI, 2>(p, nblk, dynamic, ctr, s);
        case 4: return launch_opt<EPI, 4>(p, nblk, dynamic, ctr, s);
        case 5: return launch_opt<EPI, 5>(p, nblk, dynamic, ctr, s);
        default: return launch_opt<EPI, 0>(p, nblk, dynamic, ctr, s);
    }
}

}  // namespace

extern std::atomic<int> g_gemm_pers_static;
extern int g_gemm_pers_walk;
int g_gemm_w4_opt = 0;             // A/B builds of this kernel (trace_op_set_gemm_variant(540 + opt)): bit 0 = without the re-aligning barrier, bit 2 = with the L2 touches of the A panel

// TRACE_ERR_STATE: no ticket counters for this stream and none can be made now (capturing): the caller falls back
int launch_gemm_w4(const GemmArgs& p0, int epi, hipStream_t s) {
    GemmArgs p = p0;
    p.opt = g_gemm_w4_opt;
    if (p.M < 1 || p.N % BN || p.K % BK || p.K < 3 * BK || p.fp8) return TRACE_ERR_ARG;
    if ((long)p.M * p.ldc >= (1L << 30) || (epi == EPI_RESIDUAL && (long)p.M * p.ldr >= (1L << 30))) return TRACE_ERR_ARG;   // 32-bit byte offsets
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return TRACE_ERR_ARG;                            // 32-bit piece offsets
    const int total = ((p.M + BM - 1) / BM) * (p.N / BN);
    int* ctr = nullptr;
    int nblk = 0;
    const int rc = gemm_pers_plan(s, total, &ctr, &nblk);
    if (rc != TRACE_OK) return rc;
    const int dynamic = (g_gemm_pers_static || g_gemm_pers_walk == 1) ? 0 : 1;
    bool granted = false;
    switch (epi) {
        case EPI_NONE: granted = launch_one<EPI_NONE>(p, nblk, dynamic, ctr, s); break;
        case EPI_RESIDUAL: granted = launch_one<EPI_RESIDUAL>(p, nblk, dynamic, ctr, s); break;
        case EPI_QUICKGELU: granted = launch_one<EPI_QUICKGELU>(p, nblk, dynamic, ctr, s); break;
        case EPI_SWIGLU: granted = launch_one<EPI_SWIGLU>(p, nblk, dynamic, ctr, s); break;
        default: return TRACE_ERR_ARG;
    }
    if (!granted) return TRACE_ERR_HIP;
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
