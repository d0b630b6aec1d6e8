// PERSISTENT version of gemm_ldr.hip's 256x256 loader-wave GEMM: one workgroup per CU walks output tiles, and nothing outside the
// K loop touches LDS or a barrier any more, so the loaders simply keep streaming K-tiles across the tile boundary.
//
// What gemm_ldr.hip pays per output tile outside its K loop (tools/gemm_trace.py, fc1 of the ViT: ~8 us of ~33): a fresh workgroup
// (launch, 16 address set-ups, the first K-tile's round trip = 1.6 us with nothing to compute), the accumulators staged through
// 135 KB of LDS (two workgroup barriers, every MFMA wave waiting for the slowest) and only then the stores.  Here:
//   * grid = #CUs; a workgroup takes its first tile statically and every further one from a per-XCD ticket counter (the tile order
//     stays XCD-contiguous and grouped, as in gemm.hip), fetched one tile ahead by one lane so the ticket's latency is never seen;
//   * the 4 loader waves run one continuous stream of K-tiles: the barrier that hands K-tile q to the MFMA waves lets them issue
//     K-tile q+1 — whether that is this tile's next K-tile or the NEXT tile's first one;
//   * the 8 MFMA waves pass that barrier one quarter-step EARLY (all their fragment reads of the stage are in registers by then)
//     and prefetch the next step's fragments under the last 8 MFMAs of the current one — across K-tiles and across output tiles
//     (the next tile's first fragments are read in the middle of the epilogue), so a tile starts with its operands in registers
//     and both LDS stages full;
//   * the epilogue never leaves the wave: bias (from a 512-byte LDS row the loaders fill) / QuickGELU / SwiGLU in fp32 registers,
//     packed to bf16, then ONE v_permlane16_swap per register pair turns "4 columns of one row per lane, 16 columns apart per
//     fragment" into 8 consecutive columns per lane and a DPP row_ror:8 exchange between rows r and r + 8 makes every 16-byte
//     non-temporal store instruction 8 rows x one full 128-byte line.  No LDS staging, no barrier, no waiting for the other waves:
//     the stores of tile n drain under the K loop of tile n+1 (raw s_barrier: it does not wait for the store acknowledgements);
//   * one loader wave also touches the A lines of K-tile kt + 3 into L2 (loader_role): with one K-tile in flight, load latency
//     would otherwise be the K-tile time on operands that stream from HBM.
// Same K order, same fp32 -> bf16 roundings as gemm_ldr.hip / gemm.hip: results are bit-identical (tests/test_gpu_kernels.py).
// Round 5, measured and NOT kept (profiles/r05_gemm_regstage_ab.txt): loader waves that pull K-tile q + 2 into their REGISTERS (global_load_dwordx4) and
// write it to LDS (ds_write_b128) when hand-over q + 1 has freed the stage — a K-tile period of load slack instead of none.  Bit-identical on hardware
// at the first attempt and 4-20 % SLOWER on every shape (fc1 775 vs 691 us, qkv 583 vs 489, fc2 729 vs 631, prefill gate|up 777 vs 659): with the
// operand panels in L2 the K loop is not waiting for load latency (switching the L2 touches off changes the K = 1024 shapes by < 2 %), it is short of LDS
// bandwidth — 192 KB of fragment reads + 64 KB of tile writes per K-tile against 2048 MFMA cycles — and 16 ds_write_b128 per loader wave and K-tile
// on top of that cost more than the slack buys.
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "kernels.h"
#include "gemm_tilewalk.h"

namespace {

using tilewalk::BM; using tilewalk::BN; using tilewalk::BK; using tilewalk::CTR_STRIDE;
using tilewalk::tile_coords; using tilewalk::Sched; using tilewalk::make_sched; using tilewalk::swap16;
constexpr int WM = 2, WN = 4;
constexpr int NMT = WM * WN * 64;                 // 512 MFMA threads
constexpr int NLW = 4;                            // loader waves
constexpr int NTHR = NMT + NLW * 64;              // 768
constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
constexpr int CTL_OFF = 2 * STAGE;                // int s_next[2] | 2 x 512-byte bias rows | touch scratch
constexpr int BIAS_OFF = CTL_OFF + 64;
constexpr int TOUCH_OFF = BIAS_OFF + 2 * 512;    // 4 x 256 bytes: where the loader waves' L2 touches land (never read)
constexpr int LDS_BYTES = TOUCH_OFF + 4 * 256;

__device__ __forceinline__ int swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ bf16x8_t lds16(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// One k-step (32 k of the 64-wide K-tile) of a wave's 128x64 sub-tile: 32 MFMAs.  wf / ac hold this step's W fragments and first
// two A fragments on entry and the NEXT step's on exit (NEXT): the A fragments of m-tile pairs 1..3 are read a pair ahead as in
// gemm_ldr.hip; under the last pair's 8 MFMAs the next step's first A pair and — each as soon as its last MFMA has issued — its four W
// fragments.  BARRIER: the next step reads the other LDS stage; everything this wave reads from the current stage is in registers
// before the last pair, so the barrier that releases the stage to the loaders sits there.
// LDS addresses: y is the lane's swizzled offset inside a 16-row fragment block (k-step 0; k-step 1 is y ^ 64), oa / oa_n / ow_n the
// wave-uniform offsets of the A block of this step and the A / W blocks of the next one.  y passes through an empty asm so that the
// four (lane offset x stage) combinations are NOT hoisted out of the loop as invariants: there is exactly one register for them.
// The MFMAs are inline asm with the accumulator TIED (dst = srcC = one register tuple for the whole tile).  Left to the register allocator,
// the 128 accumulators were renumbered between the peeled first / last K-tile blocks and the loop body — with 160 of 168 registers pinned it
// did that through scratch: ~20 dependent reload round trips per output tile, 12-15 us, far more than the epilogue this kernel is about.
// volatile keeps them in source order; the compiler still places the s_waitcnt for their fragment operands.  Hazards it no longer sees:
// an accumulator is touched again 32 MFMAs later (no back-to-back dependence), and the epilogue's first VALU read of one comes after an
// explicit s_nop pair.
__device__ __forceinline__ void mfma_acc(f32x4_t& c, const bf16x8_t& w, const bf16x8_t& a) {
    asm volatile("v_mfma_f32_16x16x32_" TRACE_EL " %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_new(f32x4_t& c, const bf16x8_t& w, const bf16x8_t& a) {
    asm volatile("v_mfma_f32_16x16x32_" TRACE_EL " %0, %1, %2, 0" : "=&v"(c) : "v"(w), "v"(a));
}
#define PERS_FENCE() __builtin_amdgcn_sched_barrier(0)

// OPT (A/B builds, tools/gemm_pers_ab.py): bit 1 = the hand-over barrier at the END of the k-step (fragments of the next K-tile read after it, as gemm_ldr.hip
// does) instead of before its last 8 MFMAs
template <bool FIRST, bool BARRIER, bool NEXT, int KS, int OPT>
__device__ __forceinline__ void kstep(f32x4_t (&acc)[TM][TN], bf16x8_t (&wf)[TN], bf16x8_t (&ac)[2], int& y, int oa, int oa_n, int ow_n,
                                      const char* lds) {
    constexpr bool LATE = BARRIER && (OPT & 2);
    PERS_FENCE();                                                   // every group below stays where it is written
    asm volatile("" : "+v"(y));
    const char* fa = lds + ((y ^ (KS << 6)) + oa);
    bf16x8_t an[2];
#pragma unroll
    for (int ip = 0; ip < TM / 2 - 1; ++ip) {
        an[0] = lds16(fa + (2 * ip + 2) * 2048);
        an[1] = lds16(fa + (2 * ip + 3) * 2048);
        PERS_FENCE();
        // W-fragment-major: wf[3], the last fragment the previous step requested (behind its final MFMA pair), is first read by the 7th
        // MFMA of the step instead of the 4th (OPT bit 2: the A-fragment-major order, for A/B runs)
        if (OPT & 4) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (FIRST) mfma_new(acc[2 * ip + ii][j], wf[j], ac[ii]);
                    else mfma_acc(acc[2 * ip + ii][j], wf[j], ac[ii]);
                }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    if (FIRST) mfma_new(acc[2 * ip + ii][j], wf[j], ac[ii]);
                    else mfma_acc(acc[2 * ip + ii][j], wf[j], ac[ii]);
                }
        }
        PERS_FENCE();
        ac[0] = an[0]; ac[1] = an[1];
    }
    constexpr int ip = TM / 2 - 1;
    if (BARRIER && !LATE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const char* fa_n = lds + ((y ^ ((KS ^ 1) << 6)) + oa_n);
    const char* fw_n = lds + ((y ^ ((KS ^ 1) << 6)) + ow_n);
    if (NEXT && !LATE) {
        an[0] = lds16(fa_n);
        an[1] = lds16(fa_n + 2048);
        PERS_FENCE();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (FIRST) { mfma_new(acc[2 * ip][j], wf[j], ac[0]); mfma_new(acc[2 * ip + 1][j], wf[j], ac[1]); }
        else { mfma_acc(acc[2 * ip][j], wf[j], ac[0]); mfma_acc(acc[2 * ip + 1][j], wf[j], ac[1]); }
        PERS_FENCE();
        if (NEXT && !LATE) {
            wf[j] = lds16(fw_n + j * 2048);
            PERS_FENCE();
        }
    }
    if (LATE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (NEXT) {
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = lds16(fw_n + j * 2048);
            an[0] = lds16(fa_n);
            an[1] = lds16(fa_n + 2048);
        }
    }
    if (NEXT) { ac[0] = an[0]; ac[1] = an[1]; }
    PERS_FENCE();
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// rows I0 .. I0+NI-1 (16-row m-tiles) of the wave's sub-tile: activation in registers, lane transposition, stores.  C and R are addressed
// through buffer descriptors (scalar base + ONE 32-bit byte offset register per access) whose extent is the M valid rows: rows of the
// last row panel that hang over M are dropped (stores) / read as zero (loads) by the bounds check, no predicates and no second code path.
template <int EPI, int I0, int NI>
__device__ __forceinline__ void epilogue_rows(f32x4_t (&acc)[TM][TN], uint2 (&bp)[TN], __amdgpu_buffer_rsrc_t crs, int coff, int cstep, bool hi8) {
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    constexpr int NH = GLU ? 1 : 2;               // 32-column output groups per m-tile
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        uint32_t pk[2 * NH][2];
        if (!GLU) {
            // the packed bias is re-unpacked for every m-tile (opaque to CSE): unpacked once, it is 16 registers beside 128 accumulators
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bp[j].x), "+v"(bp[j].y));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // two elements per VALU instruction where the ISA has packed fp32 (add / mul); exp2 and rcp stay one per element
                const f32x2_t b01 = {bflo(bp[j].x), bfhi(bp[j].x)}, b23 = {bflo(bp[j].y), bfhi(bp[j].y)};
                f32x2_t x01 = f32x2_t{acc[I0 + i][j][0], acc[I0 + i][j][1]} + b01;
                f32x2_t x23 = f32x2_t{acc[I0 + i][j][2], acc[I0 + i][j][3]} + b23;
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
        } else {
#pragma unroll
            for (int jj = 0; jj < TN / 2; ++jj) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gt = acc[I0 + i][2 * jj][q], up = acc[I0 + i][2 * jj + 1][q];
                    v[q] = gt * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;
                }
                pk[jj][0] = pack2bf(v[0], v[1]);
                pk[jj][1] = pack2bf(v[2], v[3]);
            }
        }
        // lane (r, g) holds columns g*4..+3 of fragment 2h in pk[2h] and of fragment 2h+1 (16 columns further) in pk[2h+1]; after the swaps it
        // holds 8 consecutive columns starting at {0, 16, 8, 24}[g] of the 32-column group h: P[h] = {pk[2h][0], pk[2h][1], pk[2h+1][0], pk[2h+1][1]}
        u32x4 P[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            swap16(pk[2 * h][0], pk[2 * h + 1][0]);
            swap16(pk[2 * h][1], pk[2 * h + 1][1]);
            P[h] = u32x4{pk[2 * h][0], pk[2 * h][1], pk[2 * h + 1][0], pk[2 * h + 1][1]};
        }
        if (NH == 2) {
            // Full 128-byte lines per store: rows r and r + 8 trade a piece (DPP row_ror:8), so that one instruction writes rows 0..7 of the m-tile
            // — lanes r < 8 their own columns 0..31, lanes r >= 8 the columns 32..63 of row r - 8 — and the next one rows 8..15.  With 64-byte
            // pieces (the two halves of a line in two instructions) the memory-side write counter read 22 % above the bytes of C.
            u32x4 A, B;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t snd = hi8 ? P[0][e] : P[NH - 1][e];
                const uint32_t rcv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)snd, 0x128, 0xf, 0xf, false);     // row_ror:8 = lane r ^ 8
                A[e] = hi8 ? rcv : P[0][e];
                B[e] = hi8 ? P[NH - 1][e] : rcv;
            }
            __builtin_amdgcn_raw_buffer_store_b128(A, crs, coff + (I0 + i) * cstep, 0, 2);                 // aux 2 = nt
            __builtin_amdgcn_raw_buffer_store_b128(B, crs, coff + (I0 + i) * cstep + (cstep >> 1), 0, 2);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(P[0], crs, coff + (I0 + i) * cstep, 0, 2);
        }
    }
}

// EPI_RESIDUAL in two passes.  Pass 1 turns the 128 accumulator registers into 64 registers of packed, lane-transposed bf16(acc + bias) —
// exactly what the one-pass epilogue holds before it adds the residual, so the roundings are unchanged — and, with the registers that
// frees, requests ALL 16 residual pieces of the lane before the first store: on this ISA loads and stores share one in-order counter, and
// a residual load issued behind a store cannot be waited for without waiting for that store's acknowledgement as well.
__device__ __forceinline__ void residual_pack(f32x4_t (&acc)[TM][TN], uint2 (&bp)[TN], u32x4 (&out)[TM][2], u32x4 (&rr)[TM][2],
                                              __amdgpu_buffer_rsrc_t rrs, int roff, int rstep, bool hi8) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bp[j].x), "+v"(bp[j].y));
        uint32_t pk[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            pk[j][0] = pack2bf(acc[i][j][0] + bflo(bp[j].x), acc[i][j][1] + bfhi(bp[j].x));
            pk[j][1] = pack2bf(acc[i][j][2] + bflo(bp[j].y), acc[i][j][3] + bfhi(bp[j].y));
        }
        u32x4 P[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            swap16(pk[2 * h][0], pk[2 * h + 1][0]);
            swap16(pk[2 * h][1], pk[2 * h + 1][1]);
            P[h] = u32x4{pk[2 * h][0], pk[2 * h][1], pk[2 * h + 1][0], pk[2 * h + 1][1]};
        }
        // the full-line arrangement of epilogue_rows (out[i][0] = rows 0..7 of the m-tile, out[i][1] = rows 8..15): the residual is read the same way
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t snd = hi8 ? P[0][e] : P[1][e];
            const uint32_t rcv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)snd, 0x128, 0xf, 0xf, false);
            out[i][0][e] = hi8 ? rcv : P[0][e];
            out[i][1][e] = hi8 ? P[1][e] : rcv;
        }
        asm volatile("" : "+v"(out[i][0]), "+v"(out[i][1]));       // the selects happen here (left lazy, P and the received pieces stay live: 96 registers for 64)
    }
    // all 16 residual pieces of the lane, before the first store (and only now: beside 128 accumulators nothing fits — hence the fence)
    PERS_FENCE();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        rr[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + i * rstep, 0, 0);
        rr[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + i * rstep + (rstep >> 1), 0, 0);
    }
}
__device__ __forceinline__ void residual_store(u32x4 (&out)[TM][2], u32x4 (&rr)[TM][2], __amdgpu_buffer_rsrc_t crs, int coff, int cstep) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 v = out[i][h];
            const u32x4 q = rr[i][h];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pack2bf(bflo(v[e]) + bflo(q[e]), bfhi(v[e]) + bfhi(q[e]));
            __builtin_amdgcn_raw_buffer_store_b128(v, crs, coff + i * cstep + h * (cstep >> 1), 0, 2);
        }
}

// ---------------- loader wave lw: pieces of 8 tile rows x 128 bytes; lw 0,1 -> A rows 0..127 / 128..255, lw 2,3 -> W ----------------
// One continuous stream of K-tiles over all the tiles of the workgroup; also publishes the next tile's index (ticket) and the tile's bias row.
template <int EPI, int OPT>
__device__ __forceinline__ void loader_role(const GemmArgs& p, const Sched& sc, char* smem, int* ctr, int dynamic, int lw, int tid, int lane) {
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    int* s_next = reinterpret_cast<int*>(smem + CTL_OFF);
    const int ntm = sc.ntm, ntn = sc.ntn, nk = sc.nk, xcd = sc.xcd, cnt = sc.cnt, base = sc.base, nwg = sc.nwg;
    const bool isA = lw < 2;
    const int half = (lw & 1) * 128;
    const int region = (isA ? 0 : A_BYTES) + half * 128;
    const char* src[16];
    int li = sc.slot, n = 0, q = 0, ticket = 0;
    uint2 bias2 = make_uint2(0u, 0u);
    // L2 touches (OPT bit 3 switches them off for A/B runs).  With two LDS stages only ONE K-tile is ever in flight, so when an operand streams
    // from HBM (ViT fc2: an 803 MB A) a K-tile costs the load's latency, not its MFMA time — and even from the Infinity Cache the pieces land
    // late often enough to show.  One byte of every A line of K-tile kt + LEAD, requested LEAD - 1 hand-overs before its LDS-DMA pieces, turns
    // those pieces into L2 hits.  ONE instruction per workgroup and K-tile: the row panel is shared by the 4 workgroups of its XCD that hold
    // column tiles tn, tn+1, .. (they run in step), so each touches the 64 rows (tn & 3) * 64 .. of it.  The touch rides BEHIND the pieces
    // of K-tile kt + 1: the in-order counter is waited down to that 1 load, not to 0.  Measured (tools/gemm_pers_ab.py, r02_gemm_pers_ab.txt,
    // vs the kernel without them): ViT fc2 shape 767 -> 571 us, out-proj shape 164 -> 158, fc1 693 -> 679, prefill qkv pair 176 -> 162, long-K
    // 325 -> 309; touching every line from every loader wave (A and W, 8 instructions per K-tile) loses on the cached shapes: the CU's
    // address path is the scarce thing; W rows split over their 8 sharers on top: mixed (-5 % on one shape, +1 % on others), left out.
    constexpr bool TOUCH = (OPT & 8) == 0;
    constexpr int LEAD = 3;
    const bool toucher = TOUCH && lw == 0;
    const char* tsrc = nullptr;                      // row (tn & 3) * 64 + lane of the tile's A panel
    // EPI_RESIDUAL: every loader wave also touches its quarter of the tile's 1024 residual lines, behind the LAST K-tile's pieces (gemm_ldr.hip)
    constexpr bool RTOUCH = TOUCH && EPI == EPI_RESIDUAL;
    int rm0 = 0, rn0 = 0;
    int pend = 0;                                    // touch instructions issued behind the latest batch of pieces
// A touch = one dword per lane by LDS-DMA into the wave's 256-byte scratch: an L2 fill with NO register destination.  (The first version used
// global_load_ubyte into a dummy register from inline asm: the compiler considers such an output dead at once and hands the register to the
// next address computation, and the load — which returns microseconds later — then overwrote a live LDS-DMA source pointer: memory faults.)
#define PERS_TOUCH_AT(PTR)                                                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTR),                                             \
                                     (__attribute__((address_space(3))) void*)(smem + TOUCH_OFF + lw * 256), 4, 0, 0)
#define PERS_TOUCH(KT) PERS_TOUCH_AT(tsrc + (KT) * 128)
#define PERS_SETUP()                                                                                                                   \
    {                                                                                                                                  \
        int tm_, tn_;                                                                                                                  \
        tile_coords(base + li, ntm, ntn, tm_, tn_);                                                                                    \
        const int m0_ = tm_ * BM, n0_ = tn_ * BN;                                                                                      \
        rm0 = m0_; rn0 = n0_;                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                                               \
            const int row = half + j * 8 + (lane >> 3);                                                                                \
            const int kc = (lane & 7) ^ ((row >> 1) & 7);                                                                              \
            src[j] = isA ? reinterpret_cast<const char*>(p.A) + ((size_t)min(m0_ + row, p.M - 1) * p.lda + kc * 8) * 2                 \
                         : reinterpret_cast<const char*>(p.W) + ((size_t)(n0_ + row) * p.ldw + kc * 8) * 2;                            \
        }                                                                                                                              \
        if (!GLU && lw == 3 && p.bias) bias2 = *reinterpret_cast<const uint2*>(p.bias + n0_ + lane * 4);  /* else stays zero */       \
        if (toucher)                                                                                                                   \
            tsrc = reinterpret_cast<const char*>(p.A) + (size_t)min(m0_ + (tn_ & 3) * 64 + lane, p.M - 1) * p.lda * 2;                 \
    }
#define PERS_ISSUE(STG, KO)                                                                                                            \
    {                                                                                                                                  \
        char* dst_ = smem + (STG) * STAGE + region;                                                                                    \
        _Pragma("unroll") for (int j = 0; j < 16; ++j)                                                                                 \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (KO)),                           \
                                             (__attribute__((address_space(3))) void*)(dst_ + j * 1024), 16, 0, 0);                    \
    }
// the wait for the latest batch of pieces: everything but the `pend` touches issued behind it
#define PERS_WAIT_PIECES()                                                                         \
    switch (pend) {                                                                                \
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;                            \
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;                            \
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;                            \
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;                            \
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;                            \
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                           \
    }
// a tile's K-tiles 1 and 2 are touched with its first pieces (K-tile 0 has no lead to gain), K-tile kt + LEAD with the pieces of kt + 1
#define PERS_TOUCH_HEAD()                                                                          \
    pend = 0;                                                                                      \
    if (toucher) {                                                                                 \
        if (1 < nk) { PERS_TOUCH(1); ++pend; }                                                  \
        if (2 < nk) { PERS_TOUCH(2); ++pend; }                                                  \
    }
    PERS_SETUP();
    PERS_ISSUE(0, 0);
    PERS_TOUCH_HEAD();
    while (true) {
        for (int kt = 0; kt < nk; ++kt, ++q) {
            PERS_WAIT_PIECES();
            if (kt == 0 && !GLU && lw == 3)                         // this tile's bias row (zeros without a bias), read by the MFMA waves after the K loop
                *reinterpret_cast<uint2*>(smem + BIAS_OFF + (n & 1) * 512 + lane * 8) = bias2;
            if (kt == 1 && tid == NMT) {                            // the tile after this one: everyone reads it after this tile's last hand-over
                s_next[(n + 1) & 1] = dynamic ? nwg + ticket : li + nwg;
                // the launch's last ticket on this XCD re-arms the counter — with an agent-scope atomic, like the tickets themselves: every draw of the
                // launch precedes it in the counter's modification order (the last draw returned cnt - 1), and it reaches memory the way the next
                // launch's draws will
                if (dynamic && ticket == cnt - 1) (void)atomicExch(ctr + xcd * CTR_STRIDE, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // K-tile q handed over; the stage of q-1 is free
            // the ticket rides with K-tile 1's pieces: from the second tile on that hand-over is an epilogue away, so its latency is free
            if (kt == 0 && dynamic && tid == NMT) ticket = atomicAdd(ctr + xcd * CTR_STRIDE, 1);
            pend = 0;
            if (kt + 1 < nk) {
                PERS_ISSUE((q + 1) & 1, (kt + 1) * 128);
                if (toucher && kt + LEAD < nk) { PERS_TOUCH(kt + LEAD); pend = 1; }
            } else {
                li = __builtin_amdgcn_readfirstlane(s_next[(n + 1) & 1]);
                const int crm0 = rm0, crn0 = rn0;                  // the tile whose last K-tile has just been handed over
                if (li < cnt) {
                    PERS_SETUP();
                    PERS_ISSUE((q + 1) & 1, 0);
                    PERS_TOUCH_HEAD();
                }
                if (RTOUCH) {
                    // residual touches of the tile just finished loading: the MFMA waves ask for R one K-tile of MFMAs + the first epilogue pass
                    // from now.  (Issued a K-tile earlier they sat in front of the s_next read above, which the compiler — seeing LDS-DMA in
                    // flight — guards with vmcnt(0): the next tile's first pieces then left ~3 us late.)
                    const int L = lw * 64 + lane;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int idx = L + t * 256, row = idx >> 2, seg = idx & 3;
                        const bf16_t* ra = p.R + (size_t)min(crm0 + row, p.M - 1) * p.ldr + crn0 + seg * 64;
                        PERS_TOUCH_AT(ra);
                    }
                    pend += 4;
                }
            }
        }
        if (li >= cnt) break;
        ++n;
    }
#undef PERS_WAIT_PIECES
#undef PERS_TOUCH_HEAD
#undef PERS_TOUCH
#undef PERS_TOUCH_AT
#undef PERS_SETUP
#undef PERS_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA (touch) of this wave may still be in flight when the workgroup's LDS is released

}

template <int EPI, int OPT>
__global__ __launch_bounds__(NTHR) void gemm_pers_kernel(GemmArgs p, int* ctr, int dynamic) {
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_next = reinterpret_cast<int*>(smem + CTL_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Sched sc = make_sched(p);
    const int ntn = sc.ntn, ntm = sc.ntm, nk = sc.nk, slot = sc.slot, cnt = sc.cnt, base = sc.base;
    if (wid >= WM * WN) {
        loader_role<EPI, OPT>(p, sc, smem, ctr, dynamic, wid - WM * WN, tid, lane);
        return;
    }

    // ---------------- MFMA waves ----------------
    const int wm = wid / WN, wn = wid % WN;
    int y = swz(lane & 15, lane >> 4);                              // lane offset inside a 16-row fragment block (rows r, 16-byte chunk g, swizzled)
    const int oa0 = wm * (BM / WM) * 128, ow0 = A_BYTES + wn * (BN / WN) * 128;      // this wave's A / W blocks inside a stage
    f32x4_t acc[TM][TN];
    bf16x8_t wf[TN], ac[2];
    int li = slot, n = 0, q = 0;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.M * p.ldc * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(EPI == EPI_RESIDUAL ? p.R : p.C), 0,
                                                                         p.M * (EPI == EPI_RESIDUAL ? p.ldr : p.ldc) * 2, 0x00020000);

#define PERS_ST(Q) (((Q) & 1) * STAGE)
#define PERS_FRAGS(Q)                                                                              \
    {                                                                                              \
        /* y is rebuilt from an id the optimiser cannot see through, so it is not kept (spilled) across the epilogue */ \
        int lane_y;                                                                                \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_y)); \
        y = swz(lane_y & 15, lane_y >> 4);                                                         \
        const char* fw_ = smem + (y + PERS_ST(Q) + ow0);                                           \
        const char* fa_ = smem + (y + PERS_ST(Q) + oa0);                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) wf[j] = lds16(fw_ + j * 2048);              \
        ac[0] = lds16(fa_);                                                                        \
        ac[1] = lds16(fa_ + 2048);                                                                 \
    }
// the two k-steps of K-tile Q; the second one hands the stage back and prefetches from K-tile Q+1 (B1 / N1)
#define PERS_KTILE(FIRST, B1, N1, Q)                                                                                               \
    kstep<FIRST, false, true, 0, OPT>(acc, wf, ac, y, PERS_ST(Q) + oa0, PERS_ST(Q) + oa0, PERS_ST(Q) + ow0, smem);                     \
    kstep<false, B1, N1, 1, OPT>(acc, wf, ac, y, PERS_ST(Q) + oa0, PERS_ST((Q) + 1) + oa0, PERS_ST((Q) + 1) + ow0, smem);
    __builtin_amdgcn_s_barrier();                                   // K-tile 0 of the first tile has landed
    PERS_FRAGS(0);
    while (true) {
        int tm, tn;
        tile_coords(base + li, ntm, ntn, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        PERS_KTILE(true, true, true, q);
        ++q;
        for (int kt = 1; kt < nk - 1; ++kt, ++q) { PERS_KTILE(false, true, true, q); }
        PERS_KTILE(false, false, false, q);
        ++q;
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");             // the last MFMAs' results are read by plain VALU code from here on
        __builtin_amdgcn_sched_barrier(0);
        const int li_next = __builtin_amdgcn_readfirstlane(s_next[(n + 1) & 1]);
        const bool has_next = li_next < cnt;
        int lane_e;
        // the lane's row / column offsets are recomputed per tile from an id the optimiser cannot hoist: as loop invariants they would
        // sit in registers across the K loop, which has none to spare (128 accumulators + 32 fragment registers of 168)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int g_e = lane_e >> 4;
        uint2 bp[TN];                                               // this lane's 4 x 4 bias values, packed (unpacked where they are used)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bp[j] = make_uint2(0u, 0u);
            if (!GLU) bp[j] = *reinterpret_cast<const uint2*>(smem + BIAS_OFF + (n & 1) * 512 + (wn * (BN / WN) + j * 16 + g_e * 4) * 2);
        }
        if (has_next) {                                             // K-tile 0 of the next tile: the loaders go on to its K-tile 1 during the epilogue
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const int pcol = ((g_e & 1) << 4) | ((g_e >> 1) << 3);      // first of the lane's 8 consecutive columns within a 32-column group
        const int mbase = m0 + wm * (BM / WM) + (lane_e & 15);
        const int cbase = (GLU ? n0 / 2 + wn * (BN / WN / 2) : n0 + wn * (BN / WN)) + pcol;
        // full-line accesses (not SwiGLU: its 32 output columns per wave are one 64-byte piece): lanes r >= 8 handle row r - 8 / r, columns 32..63
        const bool hi8 = (lane_e & 8) != 0;
        const int coff = (mbase * p.ldc + cbase) * 2 + (!GLU && hi8 ? 64 - 16 * p.ldc : 0);                                   // bytes
        if constexpr (EPI == EPI_RESIDUAL) {
            const int roff = (mbase * p.ldr + cbase) * 2 + (hi8 ? 64 - 16 * p.ldr : 0);
            u32x4 out[TM][2], rr[TM][2];
            residual_pack(acc, bp, out, rr, rrs, roff, 32 * p.ldr, hi8);
            if (has_next) PERS_FRAGS(q);
            residual_store(out, rr, crs, coff, 32 * p.ldc);
        } else {
            epilogue_rows<EPI, 0, TM / 2>(acc, bp, crs, coff, 32 * p.ldc, hi8);
            if (has_next) PERS_FRAGS(q);                            // 24 registers the first half of the epilogue has freed
            epilogue_rows<EPI, TM / 2, TM / 2>(acc, bp, crs, coff, 32 * p.ldc, hi8);
        }
        if (!has_next) break;
        li = li_next;
        ++n;
    }
#undef PERS_ST
#undef PERS_KTILE
#undef PERS_FRAGS
}

// ticket counters: 8 per (device, stream) — two launches that may run concurrently must not share them; launches on one stream are ordered.
// The CU count is cached per device (a process may hold contexts on several GPUs, each with its own default stream 0).
constexpr int MAX_DEV = 16;
int g_ncu[MAX_DEV] = {0};
std::mutex g_ctr_mu;
struct CtrKey {
    int dev; hipStream_t s;
    bool operator==(const CtrKey& o) const { return dev == o.dev && s == o.s; }
};
struct CtrHash { size_t operator()(const CtrKey& k) const { return std::hash<const void*>()((const void*)k.s) * 31u + (size_t)k.dev; } };
struct CtrState { int* ctr; int cap; };           // cap > 0: at most this many workgroups per launch on this stream (a CU-masked stream)
std::unordered_map<CtrKey, CtrState, CtrHash> g_ctrs;
// nullptr: the counters do not exist yet and cannot be made now (the stream is capturing: no allocation / memset inside a capture) or HIP failed
CtrState* counters_for(hipStream_t s, int* ncu_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_ctr_mu);
    if (!g_ncu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        g_ncu[dev] = prop.multiProcessorCount;
    }
    if (ncu_out) *ncu_out = g_ncu[dev];
    auto it = g_ctrs.find(CtrKey{dev, s});
    if (it != g_ctrs.end()) return &it->second;           // (node-based map: the address stays valid across later insertions)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    int* c = nullptr;
    if (hipMalloc(&c, 8 * CTR_STRIDE * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemsetAsync(c, 0, 8 * CTR_STRIDE * sizeof(int), s) != hipSuccess) { (void)hipFree(c); return nullptr; }   // ordered before the launch
    return &(g_ctrs[CtrKey{dev, s}] = CtrState{c, 0});
}

template <int EPI, int OPT>
void launch_opt(const GemmArgs& p, int nblk, int dynamic, int* ctr, hipStream_t s) {
    static LdsGrant grant;
    (void)grant_dynamic_lds(grant, reinterpret_cast<const void*>(gemm_pers_kernel<EPI, OPT>), LDS_BYTES);       // a refusal shows as the launch error the caller checks
    hipLaunchKernelGGL((gemm_pers_kernel<EPI, OPT>), dim3(nblk), dim3(NTHR), LDS_BYTES, s, p, ctr, dynamic);
}
int g_opt = 0;
template <int EPI>
void launch_one(const GemmArgs& p, int nblk, int dynamic, int* ctr, hipStream_t s) {
    switch (g_opt) {
        case 2: launch_opt<EPI, 2>(p, nblk, dynamic, ctr, s); break;
        case 4: launch_opt<EPI, 4>(p, nblk, dynamic, ctr, s); break;
        case 8: launch_opt<EPI, 8>(p, nblk, dynamic, ctr, s); break;
        default: launch_opt<EPI, 0>(p, nblk, dynamic, ctr, s); break;
    }
}

}  // namespace

int g_gemm_pers_opt = 0;           // A/B builds of the K loop (trace_op_set_gemm_variant(300 + opt))
std::atomic<int> g_gemm_pers_static{0};        // 1: tiles dealt round-robin instead of by ticket (A/B runs)
int g_gemm_pers_walk = 0;          // every route (trace_op_set_gemm_variant(500 + w)): 0 = tickets, atomic re-arm; 1 = static deal

int g_gemm_pers_grid_cap = 0;      // tuning knob (trace_op_set_gemm_variant(1000 + n)); streams carry their own cap: gemm_pers_set_cap.  > 0: at most this many workgroups per launch (a stream confined to part of the CUs by a CU mask: the
                                   // persistent grid must not exceed the CUs it can run on, or the surplus workgroups wait for a second round)

// Creates the ticket counters of a stream ahead of its first launch (an allocation + a memset: not something to meet inside a timed or
// captured region); launch_gemm_pers does it on demand otherwise.
int gemm_pers_init(hipStream_t s) { return counters_for(s, nullptr) ? TRACE_OK : TRACE_ERR_HIP; }
// The persistent grid of launches on stream `s` is at most `cap` workgroups (0 = the device's CU count): a stream confined to part of the CUs by a
// CU mask must not be handed more workgroups than it can run at once, or the surplus waits for a second round.  Lives with the stream's counters
// (until round 3 this was one process-wide number that outlived the streams it was set for).
int gemm_pers_set_cap(hipStream_t s, int cap) {
    if (cap < 0) return TRACE_ERR_ARG;
    CtrState* st = counters_for(s, nullptr);
    if (!st) return TRACE_ERR_HIP;
    std::lock_guard<std::mutex> lk(g_ctr_mu);       // launch_gemm_pers reads it under the same lock (the pipeline's other thread may be launching)
    st->cap = cap;
    return TRACE_OK;
}
// forgets one stream's counters (the stream is about to be destroyed and idle; a later stream may get the same handle value)
void gemm_pers_forget(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_ctr_mu);
    auto it = g_ctrs.find(CtrKey{dev, s});
    if (it != g_ctrs.end()) { (void)hipFree(it->second.ctr); g_ctrs.erase(it); }
}
// Frees the counters of every stream of `dev` (the last context on a device going away; the device must be idle)
void gemm_pers_release(int dev) {
    std::lock_guard<std::mutex> lk(g_ctr_mu);
    for (auto it = g_ctrs.begin(); it != g_ctrs.end();) {
        if (it->first.dev == dev) { (void)hipFree(it->second.ctr); it = g_ctrs.erase(it); }
        else ++it;
    }
}

// The stream's ticket counters and the persistent grid for `total` tiles, for gemm_w4.hip (launches on one stream are ordered, so the two kernels share them)
int gemm_pers_plan(hipStream_t s, int total, int** ctr, int* nblk) {
    int ncu = 0;
    CtrState* st = counters_for(s, &ncu);
    if (!st) return TRACE_ERR_STATE;
    int stream_cap;
    { std::lock_guard<std::mutex> lk(g_ctr_mu); stream_cap = st->cap; }
    const int cap = stream_cap > 0 ? stream_cap : g_gemm_pers_grid_cap;
    if (cap > 0 && cap < ncu) ncu = cap < 8 ? 8 : cap;
    *ctr = st->ctr;
    *nblk = total < ncu ? total : ncu;
    return TRACE_OK;
}

// TRACE_ERR_STATE: no ticket counters for this stream and none can be made now (capturing): the caller falls back to gemm_ldr
int launch_gemm_pers(const GemmArgs& p0, int epi, hipStream_t s) {
    GemmArgs p = p0;
    p.opt = 0;
    if (p.M < 1 || p.N % BN || p.K % BK || p.K < 2 * BK || p.fp8) return TRACE_ERR_ARG;
    if ((long)p.M * p.ldc >= (1L << 30) || (epi == EPI_RESIDUAL && (long)p.M * p.ldr >= (1L << 30))) return TRACE_ERR_ARG;   // 32-bit byte offsets
    int ncu = 0;
    CtrState* st = counters_for(s, &ncu);
    if (!st) return TRACE_ERR_STATE;
    int* ctr = st->ctr;
    int stream_cap;
    { std::lock_guard<std::mutex> lk(g_ctr_mu); stream_cap = st->cap; }
    const int cap = stream_cap > 0 ? stream_cap : g_gemm_pers_grid_cap;      // the stream's own cap (CU-masked streams), else the process-wide tuning knob
    if (cap > 0 && cap < ncu) ncu = cap < 8 ? 8 : cap;                 // every XCD keeps a workgroup: tiles are dealt per XCD
    const int total = ((p.M + BM - 1) / BM) * (p.N / BN);
    // g_gemm_pers_static == 2: one workgroup per tile (the dispatcher places them as CUs free up, nothing persists): this kernel's K loop and
    // register epilogue without the tile walk (A/B runs)
    const int nblk = g_gemm_pers_static == 2 ? total : (total < ncu ? total : ncu);
    const int dynamic = (g_gemm_pers_static || g_gemm_pers_walk == 1) ? 0 : 1;
    g_opt = g_gemm_pers_opt;
    switch (epi) {
        case EPI_NONE: launch_one<EPI_NONE>(p, nblk, dynamic, ctr, s); break;
        case EPI_RESIDUAL: launch_one<EPI_RESIDUAL>(p, nblk, dynamic, ctr, s); break;
        case EPI_QUICKGELU: launch_one<EPI_QUICKGELU>(p, nblk, dynamic, ctr, s); break;
        case EPI_SWIGLU: launch_one<EPI_SWIGLU>(p, nblk, dynamic, ctr, s); break;
        default: return TRACE_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
