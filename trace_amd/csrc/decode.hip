// Decode-step kernels (1 new token for each of B <= 64 sequences).  This is the HBM-bound heart of the path:
// every step streams all 7.2 B bf16 weights once plus the KV cache of every sequence (reference:
// TraceMistralForCausalLM.forward with input_ids [B,1] + past_key_values, trace/model/language_model/trace_mistral.py:114-264).
//
// skinny_lds:   out[b,n] = sum_k X[b,k] W[n,k] on the 16x16x32 bf16 MFMA with the weights as the A operand, streamed
//   once from a tile-contiguous decode copy with non-temporal 16-byte loads, and the activations as B, parked in
//   LDS.  The k-slot permutation trick (A and B fragments only have to agree on which k each slot means) lets each
//   lane read 32 contiguous bytes.  The four GEMVs of a layer leave fp32 k-chunk partial rows; their consumers
//   (attn_decode, add_rmsnorm, swiglu_combine) sum them on load.
// attn_decode:  single-query GQA attention over the KV cache on the matrix cores (V cache stored transposed), split
//   over the context, + RoPE, cache append and the cross-split merge.
// head_logits / select_next:  active-head GEMV, masked arg-max (trace_mistral.py:244-252 + HF greedy), the
//   head-switch state machine (trace_mistral.py:86-88,336-344) and the next-token embedding
//   (trace_arch.py:345-375) — all on device, so a decode step never returns to the host.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 ldg_nt(const bf16_t* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
union Frag { uint4 u; bf16x8_t v; };

// ---------------------------------------------------------------------------------------------------------
// skinny_lds: decode GEMV with the ACTIVATIONS STATIONARY IN LDS.
// History: the first version gave every workgroup 16 weight rows and all of K, and read X (B x K bf16, L2-resident)
// straight into MFMA fragments.  At B = 32 that L2 stream is as large as the weight stream itself and cost 30-60 %
// (gate|up 72 us vs 55 us with the X loads neutered, down 45 vs 29, qkv 26 vs 16); its 16 interleaved 8 KB-strided
// row streams per wave also thrashed DRAM pages (the same bytes read as one contiguous run: 51.7 -> 41.7 us).
// Now one workgroup per CU parks a K-chunk of X (<= 128 KB: 4096 k for 16 rows, 2048 k for 32, 1024 k for 64) in LDS once, in
// MFMA-fragment order (every later read is a lane-linear ds_read_b128), and streams several weight tiles against it:
//   grid = KS k-chunks x row-groups (<= #CUs workgroups), workgroup = T <= 16 tasks on <= 8 waves; a task is NT 16-row
//   weight tiles over the chunk; when T <= 4 its k-units are split over WPT waves (LDS-reduced, fixed order).
// Weights: `tiled` = the decode copy [N/16][K/64][64 lanes][16] (one 2 KB block per 16 rows x 64 k, lane-linear, so
// a wave's stream is ONE contiguous run); row-major [N][K] is kept for small callers (STC squeeze-excite).
// Epilogues: EPI_PARTIAL (the decode step's path) stores plain fp32 partial rows [ks][SK_ROWS][N] and the CONSUMER kernel
// sums the chunks on load.  EPI_NONE / RESIDUAL / SWIGLU finish in-kernel: with KS > 1 the chunk partials of a tile
// ([KS][NT*NB][64 lanes][4] fp32, written through) are merged by whichever wave takes the tile's last agent-scope
// ticket, always in chunk order, so results do not depend on arrival order — but that merge costs 5-8 us of dependent
// round trips (store drain -> ticket -> acquire -> loads) per launch, which is why the decode step does not use it.
// PRO (round 3, batches of at most SKINNY_PRO_ROWS): the GEMV takes its activations from the PREVIOUS GEMV's fp32 partial rows instead of a bf16
// matrix and does the decode step's "sum the chunks + residual -> new residual, RMSNorm" itself while it parks them — what add_rmsnorm_kernel
// does between two GEMVs (4.5 us + a kernel boundary, twice per layer, of a 109 us layer at batch 1).  Every workgroup needs sum x^2 over the
// WHOLE row, so every workgroup redoes the row's sum (B x K x (ks + 1) loads from L2: nothing at 1-4 rows, the reason it stops there), keeps its
// groups in registers, writes the normalised bf16 values of ITS K-chunk into LDS in fragment order (the image the LDS-DMA parking builds) and,
// if it is row-group 0, the chunk's new residual values to xout (!= R: other workgroups are still reading R).  Same sums, same roundings as
// add_rmsnorm_kernel; only the order of the sum of squares differs (block reduction).
struct SkinnyPro {
    const float* part; int ks;     // fp32 partial rows [ks][SK_ROWS][K] of the previous GEMV (ks == 0: none, x = R)
    const bf16_t* R; int ldr;      // residual rows [B][K]
    bf16_t* xout; int ldx;         // new residual rows (ping-pong partner of R)
    const bf16_t* w; float eps;    // RMSNorm weight [K]
};
constexpr int SKINNY_PRO_ROWS = 4;

template <int EPI, int NB, int NT, int PRO = 0>
__global__ __launch_bounds__(512) void skinny_lds_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ W, int ldw,
                                                         bf16_t* __restrict__ out, int ldo, const bf16_t* __restrict__ R, int ldr,
                                                         int B, int K, int chunk_units, int KS, int T, int WPT, int ntiles,
                                                         float* __restrict__ ws, unsigned int* __restrict__ tickets, int tiled,
                                                         int dbg, SkinnyPro pro) {
    // NT = 16-row weight tiles per task (2: a gate|up pair, or two neighbouring tiles of a wide PARTIAL product)
    constexpr int UN = (NT == 2) ? 2 : 4;                // 64-wide k units per load batch: 8 KB of weights per batch per wave
    constexpr int NF = NT * NB;                          // accumulator fragments per task
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);                                      // [unit][half][nb][lane]
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + (size_t)chunk_units * 2 * NB * 1024); // [task][wsub][NF][lane]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int ks = blockIdx.x % KS, rg = blockIdx.x / KS;
    const int U = K >> 6;
    const int u_beg = ks * chunk_units, nu = min(U - u_beg, chunk_units);
    const int team = wid / WPT, wsub = wid - team * WPT, nteams = nwaves / WPT;
    const int ua = (wsub * nu) / WPT, ub = ((wsub + 1) * nu) / WPT;                     // this wave's units of the chunk
    const int ustride = tiled ? 1024 : 64;
    const bf16_t* wp[NT];
    Frag wa[UN][NT][2], wb[UN][NT][2];
    auto loadw = [&](Frag (&wf)[UN][NT][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const bool ok = u + j < ub;
            const int ko = (u + j) * ustride;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wf[j][t][0].u = ok ? ldg_nt(wp[t] + ko) : make_uint4(0, 0, 0, 0);
                wf[j][t][1].u = ok ? ldg_nt(wp[t] + ko + 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    f32x4_t acc[NT][NB];
    // a team takes tasks team, team + nteams, ... of the workgroup's T (WPT > 1: nteams == T, one pass for every wave,
    // so the barriers below stay workgroup-uniform)
    bool parked = false;
    for (int task = team; task < T; task += nteams) {
    const int tile = rg * T + task;
    const bool active = tile < ntiles;
    const int n0 = tile * 16 * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (tiled) wp[t] = W + ((size_t)(active ? tile * NT + t : 0) * U + u_beg) * 1024 + lane * 16;
        else wp[t] = W + (size_t)(active ? n0 + t * 16 + r : 0) * ldw + (size_t)u_beg * 64 + g * 16;
    }
    const bool work = active && ua < ub;
    if (work) loadw(wa, ua);                             // the weight stream starts before the activations are parked

    // ---- park X[:, chunk] in LDS in fragment order: combo c = (unit*2 + half)*NB + nb, lane (r, g) holds
    //      X[16 nb + r][(u_beg + unit)*64 + g*16 + half*8 .. +8]  (zeros for rows >= B) ----
    if constexpr (PRO == 1) {
        if (!parked) {
            __shared__ float s_ss[SKINNY_PRO_ROWS][8];
            const int nthr = blockDim.x, ngrp = K >> 3;          // 8-element groups per row
            constexpr int MAXG = 2;                               // 8-element groups per thread and row (the launcher checks K <= 16 * threads)
            uint4 xv[SKINNY_PRO_ROWS][MAXG];
            float ss[SKINNY_PRO_ROWS];
#pragma unroll
            for (int b = 0; b < SKINNY_PRO_ROWS; ++b) {
                ss[b] = 0.f;
                if (b < B) {
#pragma unroll
                    for (int q = 0; q < MAXG; ++q) {
                        const int e8 = tid + q * nthr;
                        xv[b][q] = make_uint4(0u, 0u, 0u, 0u);
                        if (e8 < ngrp) {
                            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                            const float* pp = pro.part + (size_t)b * K + e8 * 8;
                            for (int k2 = 0; k2 < pro.ks; ++k2) {      // chunk order
                                a0 += *reinterpret_cast<const f32x4_t*>(pp + (size_t)k2 * SK_ROWS * K);
                                a1 += *reinterpret_cast<const f32x4_t*>(pp + (size_t)k2 * SK_ROWS * K + 4);
                            }
                            const uint4 rr = *reinterpret_cast<const uint4*>(pro.R + (size_t)b * pro.ldr + e8 * 8);
                            uint4 xo = rr;
                            if (pro.ks > 0) {                       // bf16(sum) + residual, rounded: add_rmsnorm_kernel's arithmetic
                                xo.x = pack2bf(bf2f(f2bf(a0[0])) + bflo(rr.x), bf2f(f2bf(a0[1])) + bfhi(rr.x));
                                xo.y = pack2bf(bf2f(f2bf(a0[2])) + bflo(rr.y), bf2f(f2bf(a0[3])) + bfhi(rr.y));
                                xo.z = pack2bf(bf2f(f2bf(a1[0])) + bflo(rr.z), bf2f(f2bf(a1[1])) + bfhi(rr.z));
                                xo.w = pack2bf(bf2f(f2bf(a1[2])) + bflo(rr.w), bf2f(f2bf(a1[3])) + bfhi(rr.w));
                            }
                            xv[b][q] = xo;
                            const uint32_t u[4] = {xo.x, xo.y, xo.z, xo.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) ss[b] = fmaf(bflo(u[e]), bflo(u[e]), fmaf(bfhi(u[e]), bfhi(u[e]), ss[b]));
                            const int unit = e8 >> 3;
                            if (rg == 0 && unit >= u_beg && unit < u_beg + nu)       // this chunk's share of the new residual row, written once
                                *reinterpret_cast<uint4*>(pro.xout + (size_t)b * pro.ldx + e8 * 8) = xo;
                        }
                    }
                    ss[b] = wave_sum(ss[b]);
                    if (lane == 0) s_ss[b][wid] = ss[b];
                }
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < SKINNY_PRO_ROWS; ++b) {
                if (b < B) {
                    float tot = 0.f;
                    for (int w2 = 0; w2 < nwaves; ++w2) tot += s_ss[b][w2];
                    const float rstd = rsqrtf(tot / (float)K + pro.eps);
#pragma unroll
                    for (int q = 0; q < MAXG; ++q) {
                        const int e8 = tid + q * nthr, unit = e8 >> 3;
                        if (e8 < ngrp && unit >= u_beg && unit < u_beg + nu) {
                            const uint4 wv = *reinterpret_cast<const uint4*>(pro.w + e8 * 8);
                            const uint32_t xu[4] = {xv[b][q].x, xv[b][q].y, xv[b][q].z, xv[b][q].w}, wu[4] = {wv.x, wv.y, wv.z, wv.w};
                            u32x4_t y;
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = pack2bf(bflo(xu[e]) * rstd * bflo(wu[e]), bfhi(xu[e]) * rstd * bfhi(wu[e]));
                            // fragment order: combo (unit, half) -> 1 KB image, lane (r = b, g): X[b][unit*64 + g*16 + half*8 .. +8]
                            const int g2 = (e8 & 7) >> 1, half = e8 & 1;
                            xs[(((unit - u_beg) * 2 + half) * NB) * 64 + g2 * 16 + b] = y;
                        }
                    }
                }
            }
            __syncthreads();
            parked = true;
        }
    }
    if constexpr (PRO == 2) {
        // SwiGLU prologue (round 4, batch 1 .. SKINNY_PRO_ROWS): this GEMV (down) takes its activations from the gate|up GEMV's fp32 partial rows
        // [pro.ks][SK_ROWS][2 K] (16-row interleaved: columns [32p, 32p+16) gate, [32p+16, 32p+32) up of activation columns [16p, 16p+16)) and does
        // swiglu_combine_kernel's work for ITS K-chunk while it parks: sums in chunk order, silu(gate) * up in fp32, one bf16 rounding — the same
        // arithmetic, so the products are bit-identical to the two-kernel form; one launch and one kernel boundary less per layer.  The row groups
        // of a chunk each redo it (B x chunk x 2 x ks floats from L2: 15 MB per launch at one row).
        if (!parked) {
            const int ngrp = nu * 8, N2 = 2 * K;
            for (int idx = tid; idx < B * ngrp; idx += blockDim.x) {
                const int b = idx / ngrp, e8l = idx - b * ngrp, e8 = u_beg * 8 + e8l;
                const float* base = pro.part + (size_t)b * N2 + (size_t)(e8 >> 1) * 32 + (e8 & 1) * 8;
                f32x4_t g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f}, u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
                for (int k2 = 0; k2 < pro.ks; ++k2) {          // chunk order
                    const float* q = base + (size_t)k2 * SK_ROWS * N2;
                    g0 += *reinterpret_cast<const f32x4_t*>(q);
                    g1 += *reinterpret_cast<const f32x4_t*>(q + 4);
                    u0 += *reinterpret_cast<const f32x4_t*>(q + 16);
                    u1 += *reinterpret_cast<const f32x4_t*>(q + 20);
                }
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = g0[e] / (1.f + __expf(-g0[e])) * u0[e];
                    o[4 + e] = g1[e] / (1.f + __expf(-g1[e])) * u1[e];
                }
                const u32x4_t y = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
                // fragment order, as in the RMSNorm prologue: combo (unit, half) -> 1 KB image, lane (r = b, g): X[b][unit*64 + g*16 + half*8 .. +8]
                const int unit = e8l >> 3, g2 = (e8l & 7) >> 1, half = e8l & 1;
                xs[((unit * 2 + half) * NB) * 64 + g2 * 16 + b] = y;
            }
            __syncthreads();
            parked = true;
        }
    }
    if (!parked && dbg != 6) {                          // (dbg == 6: the register-staged parking below, for A/B runs)
        // parking by LDS-DMA: a combo is one lane-linear 1 KB image whose lanes read arbitrary 16-byte sources — exactly what
        // global_load_lds does, with no register round trip, no ds_write and no index math in the way of the weight stream
        // (rows >= B read row B-1 instead of zeros: their accumulator rows are never stored)
        const int combos = nu * 2 * NB;
        for (int c = wid; c < combos; c += nwaves) {
            const int nb = c % NB, uh = c / NB, h = uh & 1, u = uh >> 1;
            const int m = min(16 * nb + r, B - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)m * ldx + (size_t)(u_beg + u) * 64 + g * 16 + h * 8),
                                             (__attribute__((address_space(3))) void*)(smem + (size_t)c * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        parked = true;
    }
    if (!parked) {
        const int combos = nu * 2 * NB;
        for (int c0 = wid; c0 < combos; c0 += nwaves * 4) {
            u32x4_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * nwaves;
                const int nb = c % NB, uh = c / NB, h = uh & 1, u = uh >> 1;
                const int m = 16 * nb + r;
                v[q] = u32x4_t{0u, 0u, 0u, 0u};
                if (c < combos && m < B)
                    v[q] = *reinterpret_cast<const u32x4_t*>(X + (size_t)m * ldx + (size_t)(u_beg + u) * 64 + g * 16 + h * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * nwaves;
                if (c < combos) xs[c * 64 + lane] = v[q];
            }
        }
        __syncthreads();
        parked = true;
    }

#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](Frag (&wf)[UN][NT][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            if (u + j < ub) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8_t x0 = __builtin_bit_cast(bf16x8_t, xs[(((u + j) * 2 + 0) * NB + nb) * 64 + lane]);
                    const bf16x8_t x1 = __builtin_bit_cast(bf16x8_t, xs[(((u + j) * 2 + 1) * NB + nb) * 64 + lane]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t][nb] = mfma16(wf[j][t][0].v, x0, acc[t][nb]);
                        acc[t][nb] = mfma16(wf[j][t][1].v, x1, acc[t][nb]);
                    }
                }
            }
        }
    };
    if (work) {
        for (int u = ua; u < ub; u += 2 * UN) {
            if (u + UN < ub) loadw(wb, u + UN);
            mma(wa, u);
            if (u + UN < ub) {
                if (u + 2 * UN < ub) loadw(wa, u + 2 * UN);
                mma(wb, u + UN);
            }
        }
    }
    // ---- the WPT waves of a task: fixed-order sum through LDS (workgroup-uniform branch) ----
    if (WPT > 1) {
        if (active) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) red[((team * WPT + wsub) * NF + t * NB + nb) * 64 + lane] = acc[t][nb];
        }
        __syncthreads();
        if (active && wsub == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4_t sacc = red[((team * WPT) * NF + t * NB + nb) * 64 + lane];
                    for (int w = 1; w < WPT; ++w) sacc += red[((team * WPT + w) * NF + t * NB + nb) * 64 + lane];
                    acc[t][nb] = sacc;
                }
        }
    }
    if (!active || wsub != 0 || dbg == 3) continue;
    if (EPI == EPI_PARTIAL) {                           // fp32 partial rows [ks][SK_ROWS][N = ldo]; the consumer sums the chunks
        float* pr = ws + (size_t)ks * SK_ROWS * ldo;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int m = 16 * nb + r;
            if (m < B) {
#pragma unroll
                for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4_t*>(pr + (size_t)m * ldo + n0 + t * 16 + g * 4) = acc[t][nb];
            }
        }
        continue;
    }
    // ---- K-chunk partials: publish, ticket, the last wave of the tile merges in chunk order ----
    if (KS > 1) {
        float* wt = ws + (size_t)tile * KS * NF * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 4; ++e)       // write-through (sc1) stores: at agent scope once vmcnt drains
                    __hip_atomic_store(&wt[((size_t)(ks * NF + t * NB + nb) * 4 + e) * 64 + lane], acc[t][nb][e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dbg == 4) continue;
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(&tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk != (unsigned)(KS - 1)) continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int k2 = 0; k2 < KS; ++k2) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][nb][e] += wt[((size_t)(k2 * NF + t * NB + nb) * 4 + e) * 64 + lane];
        }
        if (lane == 0) __hip_atomic_store(&tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- epilogue straight from the accumulators: lane (r, g) holds out[m = 16 nb + r][n0 + t*16 + g*4 + 0..3] ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int m = 16 * nb + r;
        if (m >= B) continue;
        float o[4];
        if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gt = acc[0][nb][e], up = acc[NT - 1][nb][e];
                o[e] = gt / (1.f + __expf(-gt)) * up;
            }
            *reinterpret_cast<uint2*>(out + (size_t)m * ldo + (n0 >> 1) + g * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[0][nb][e];
            if (EPI == EPI_RESIDUAL) {
                const uint2 rr = *reinterpret_cast<const uint2*>(R + (size_t)m * ldr + n0 + g * 4);
                o[0] = bf2f(f2bf(o[0])) + bflo(rr.x); o[1] = bf2f(f2bf(o[1])) + bfhi(rr.x);
                o[2] = bf2f(f2bf(o[2])) + bflo(rr.y); o[3] = bf2f(f2bf(o[3])) + bfhi(rr.y);
            }
            *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n0 + g * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
    }
    }   // task loop
}

// SwiGLU over EPI_PARTIAL rows of the gate|up product (16-row interleaved: columns [32p, 32p+16) gate, [32p+16, 32p+32)
// up of output columns [16p, 16p+16)): out[b][j] = bf16(silu(sum_ks gate) * sum_ks up).  4 outputs per thread.
__global__ __launch_bounds__(256) void swiglu_combine_kernel(const float* __restrict__ part, int KS, int N2, bf16_t* __restrict__ out,
                                                             int ldo, int B) {
    const int I4 = N2 >> 3;                              // float4 groups per output row
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * I4) return;
    const int b = idx / I4, j4 = idx - b * I4;
    const int p = j4 >> 2, i = (j4 & 3) * 4;
    const float* base = part + (size_t)b * N2 + p * 32 + i;
    f32x4_t gt = {0.f, 0.f, 0.f, 0.f}, up = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KS; k0 += 4) {              // chunk rows fetched four at a time, summed in chunk order
        f32x4_t tg[4], tu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = k0 + u < KS;
            tg[u] = ok ? *reinterpret_cast<const f32x4_t*>(base + (size_t)(k0 + u) * SK_ROWS * N2) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            tu[u] = ok ? *reinterpret_cast<const f32x4_t*>(base + (size_t)(k0 + u) * SK_ROWS * N2 + 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { gt += tg[u]; up += tu[u]; }
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gt[e] / (1.f + __expf(-gt[e])) * up[e];
    *reinterpret_cast<uint2*>(out + (size_t)b * ldo + j4 * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
}

// Row-major W [N][K] -> the decode copy [N/16][K/64][64 lanes][16]: lane (r, g) of block (tile, unit) holds
// W[tile*16 + r][unit*64 + g*16 .. +16] (its two MFMA fragments back to back).  One thread per 16-byte piece.
__global__ __launch_bounds__(256) void tile_pack_kernel(const bf16_t* __restrict__ src, int ldw, bf16_t* __restrict__ dst, int N, int K) {
    const int U = K >> 6;
    const long total = (long)N * (K >> 3);
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int n = (int)(p / (K >> 3)), k8 = (int)(p - (long)n * (K >> 3));
        const int tile = n >> 4, r = n & 15, u = k8 >> 3, g = (k8 & 7) >> 1, h = k8 & 1;
        *reinterpret_cast<u32x4_t*>(dst + (((size_t)tile * U + u) * 64 + g * 16 + r) * 16 + h * 8) =
            *reinterpret_cast<const u32x4_t*>(src + (size_t)n * ldw + (size_t)k8 * 8);
    }
}

// Decode residual add + RMSNorm fed by EPI_PARTIAL: x = bf16(sum_ks part[ks][b][:]) + R[b][:] (the GEMV epilogue's
// rounding), stored as the new residual stream, then y = RMSNorm(x) * w.  One 1024-thread workgroup per sequence, 4
// elements per thread; the k-chunk rows are fetched eight at a time (a plain accumulate loop serialises one HBM round
// trip per chunk: 13 us at 16 chunks) and summed in chunk order.
__global__ __launch_bounds__(1024) void add_rmsnorm_kernel(const float* __restrict__ part, int KS, const bf16_t* __restrict__ R, int ldr,
                                                           bf16_t* __restrict__ xout, int ldx, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, int ldy, int N, float eps,
                                                           uint8_t* __restrict__ y8, float* __restrict__ sy) {
    __shared__ float s_red[16];
    __shared__ float s_amax[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int c = tid;                                 // N <= 4096: one 4-element chunk per thread
    const bool on = c < (N >> 2);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    uint2 wv = make_uint2(0u, 0u);
    float ss = 0.f;
    if (on) {
        wv = *reinterpret_cast<const uint2*>(w + c * 4);
        const uint2 rr = *reinterpret_cast<const uint2*>(R + (size_t)b * ldr + c * 4);
        const float* p0 = part + (size_t)b * N + c * 4;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < KS; k0 += 8) {
            f32x4_t t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = k0 + u < KS ? *reinterpret_cast<const f32x4_t*>(p0 + (size_t)(k0 + u) * SK_ROWS * N) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) a += t[u];
        }
        const float x0 = bf2f(f2bf(a[0])) + bflo(rr.x), x1 = bf2f(f2bf(a[1])) + bfhi(rr.x);
        const float x2 = bf2f(f2bf(a[2])) + bflo(rr.y), x3 = bf2f(f2bf(a[3])) + bfhi(rr.y);
        const uint2 xo = make_uint2(pack2bf(x0, x1), pack2bf(x2, x3));
        *reinterpret_cast<uint2*>(xout + (size_t)b * ldx + c * 4) = xo;
        v[0] = bflo(xo.x); v[1] = bfhi(xo.x); v[2] = bflo(xo.y); v[3] = bfhi(xo.y);
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += v[e] * v[e];
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += s_red[i];
    const float rstd = rsqrtf(tot / (float)N + eps);
    if (on) {
        const float o0 = v[0] * rstd * bflo(wv.x), o1 = v[1] * rstd * bfhi(wv.x);
        const float o2 = v[2] * rstd * bflo(wv.y), o3 = v[3] * rstd * bfhi(wv.y);
        const uint2 yo = make_uint2(pack2bf(o0, o1), pack2bf(o2, o3));
        *reinterpret_cast<uint2*>(y + (size_t)b * ldy + c * 4) = yo;
        v[0] = bflo(yo.x); v[1] = bfhi(yo.x); v[2] = bflo(yo.y); v[3] = bfhi(yo.y);
    }
    if (y8) {
        // fp8 weight path: the next GEMV's activations, quantised here exactly as fp8.hip's quant_rows_fp8 would from y (per-row amax of
        // the bf16-rounded values, q = rne(y * 448 / amax)) — saves that launch
        float am = on ? fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) : 0.f;
        am = wave_max(am);
        if ((tid & 63) == 0) s_amax[tid >> 6] = am;
        __syncthreads();
        am = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) am = fmaxf(am, s_amax[i]);
        const float inv = am > 0.f ? 448.f / am : 1.f;
        if (tid == 0) sy[b] = am > 0.f ? am / 448.f : 1.f;
        if (on) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf(v[e] * inv, -448.f), 448.f);
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
            *reinterpret_cast<int*>(y8 + (size_t)b * N + c * 4) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Single-query GQA attention over the KV cache, one launch per layer: RoPE of q / the new k, the KV-cache
// append, the split-context partial attention and the cross-split combine are all in this kernel.
//
// Cache layout: K row-major [slot][kvh][pos][128]; V TRANSPOSED [slot][kvh][128][ctx_stride] so that both
// products run on the matrix cores with every lane streaming contiguous 16-byte pieces:
//   S^T[pos, head]  = K[pos, :] . q[head, :]      A = 16 cache rows (lane: 2 x 16 B of one row), B = the 4 q heads
//                                                 of this kv group (columns 4..15 are zero)
//   O^T[d, head]   += V^T[d, pos] . P^T[pos, head] A = 16 rows of V^T (lane: 8 consecutive positions), B = P
// A wave iteration covers 32 positions.  The A-row -> position map of the two S tiles is chosen so that the 8 scores
// a lane ends up holding (2 tiles x 4 accumulator registers) are 8 CONSECUTIVE positions — exactly the k-slots the
// second MFMA wants for that lane — so P never leaves registers (the swapped-operand flash-attention trick).
// (The previous VALU formulation spent ~200 VALU ops per cache row per lane and capped at 3.3 TB/s of cache stream.)
//
// grid (nsplit, nkv, B), 256 threads = 4 waves, wave w takes 32-position blocks w, w+4, ... of the split's chunk
// (16 KB of K + V^T per block, re-requested as soon as the registers are free).  Partials go to `ws`
// ([b][q-head][split][hd + 2] fp32) with write-through (sc1) stores; every wave drains vmcnt, then one lane takes
// an agent-scope ticket; the last-arriving workgroup of a (b, kv-head) pair does ONE agent-scope acquire and merges
// the splits (placement-independent: no assumption on dispatch order or XCD), then re-zeroes the ticket.
// Masked positions contribute p = 0 times whatever the cache holds there: the caches are zero-initialised and only
// ever hold finite values.
// NW = waves per workgroup (round 6): 4, or 3 when the launch has more workgroups than 3 x #CUs can hold of the 4-wave form — 128 sequences x 8 kv heads =
// 1024 workgroups against 768 resident slots ran as 1.33 rounds with a tail of one workgroup per CU; as 3-wave workgroups four fit a CU (the same 12
// waves) and all 1024 are resident from the start.  NT = HBM loads of the cache with the non-temporal hint (A/B).
template <int NW, bool NT>
__global__ __launch_bounds__(NW * 64, 3) void attn_decode_kernel(const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ kcache,
                                                             bf16_t* __restrict__ vtcache, long slot_stride, long kv_head_stride,
                                                             int ctx_stride, const int32_t* __restrict__ slots,
                                                             const int32_t* __restrict__ pos, float* __restrict__ ws,
                                                             unsigned int* __restrict__ tickets, bf16_t* __restrict__ O, int ldo,
                                                             int nq, int nkv, int nsplit, float scale, int fuse_rope,
                                                             const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                             const float* __restrict__ qpart, int qks, int dbg) {
    constexpr int HD = 128, GQ = 4;
    __shared__ __attribute__((aligned(16))) float s_acc[NW][GQ][HD];
    __shared__ float s_m[NW][GQ], s_l[NW][GQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_new[2 * HD];   // roped k | v of the newest position (owner split)
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i = lane & 15, g = lane >> 4;          // i: A-row / head column; g: k-group
    const int sp = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int p_new = pos[b];
    const int ctx = p_new + 1;
    int chunk = (ctx + nsplit - 1) / nsplit;
    chunk = (chunk + 31) & ~31;
    const int beg = sp * chunk, end = min(ctx, beg + chunk);
    const int len = max(end - beg, 0);
    const int nit = (len + 31) >> 5;                  // 32-position blocks in this split
    bf16_t* kb = kcache + (size_t)slots[b] * slot_stride + (size_t)kvh * kv_head_stride;
    bf16_t* vb = vtcache + (size_t)slots[b] * slot_stride + (size_t)kvh * kv_head_stride;
    // the qkv row of this sequence: bf16 [ldq], or (qpart) the qkv GEMV's fp32 k-chunk partial rows [qks][SK_ROWS][ldq], summed
    // and rounded to bf16 here (what the GEMV epilogue would have stored)
    const bf16_t* row = qkv + (size_t)b * ldq;
    auto slice8 = [&](int n) -> u32x4_t {
        if (!qpart) return *reinterpret_cast<const u32x4_t*>(row + n);
        const float* pp = qpart + (size_t)b * ldq + n;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < qks; k0 += 4) {          // chunk rows fetched four at a time, summed in chunk order
            f32x4_t ta[4], tc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = k0 + u < qks;
                ta[u] = ok ? *reinterpret_cast<const f32x4_t*>(pp + (size_t)(k0 + u) * SK_ROWS * ldq) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                tc[u] = ok ? *reinterpret_cast<const f32x4_t*>(pp + (size_t)(k0 + u) * SK_ROWS * ldq + 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += ta[u]; c += tc[u]; }
        }
        return u32x4_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(c[0], c[1]), pack2bf(c[2], c[3])};
    };
    auto elem = [&](int n) -> bf16_t {
        if (!qpart) return row[n];
        float a = qpart[(size_t)b * ldq + n];
        for (int k2 = 1; k2 < qks; ++k2) a += qpart[((size_t)k2 * SK_ROWS + b) * ldq + n];
        return f2bf(a);
    };
    const bool owner = fuse_rope && len > 0 && end == ctx;       // this split holds the newest position

    auto ldc = [](const bf16_t* p) -> u32x4_t {        // one 16-byte piece of the cache
        if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        else return *reinterpret_cast<const u32x4_t*>(p);
    };
    const int prow = (i >> 2) * 8 + (i & 3);          // + 4 t: position (within the block) of A-row i of S tile t
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    u32x4_t kr[2][4], vr[8];
    // The loads are UNCONDITIONAL (round 6): a block's 32 positions lie below round_up(end, 32) <= ctx_stride, i.e. inside the slot's rows, and what a position >= end
    // (or the newest row, spliced from LDS below) holds never counts — its score is masked, its p is 0, whole groups of 8 past `end` are zeroed before the PV
    // product of a wave's last block, and the caches hold finite values only (see above).  With
    // per-lane predicates the loads sat in exec-masked branches, the number in flight was unknown at the joins, and hipcc put s_waitcnt vmcnt(0) in front of the
    // QK MFMAs: every wave waited for the V^T block it had requested a moment earlier instead of computing S under it (ISA read, round 6; measured against the
    // predicated form in profiles/r06_decode_attn_waits_ab.txt: equal at batch 128, where 12 waves per CU cover the latency anyway, -1.4 % per step at batch 16).
    auto load_k = [&](int it) {
        const int P0 = beg + it * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = P0 + prow + 4 * t;
            const bf16_t* src = kb + (size_t)p * HD + g * 16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) kr[t][s4] = ldc(src + (s4 >> 1) * 64 + (s4 & 1) * 8);
        }
    };
    auto load_v = [&](int it) {
        const int P0 = beg + it * 32;
        // (dbg == 8, timing only: the same 8 KB read as ONE contiguous block [128 d][32 positions] — what a position-blocked V^T cache would give)
        const bf16_t* vsrc = dbg == 8 ? vb + ((size_t)(P0 >> 5) * HD + i) * 32 + g * 8 : vb + (size_t)i * ctx_stride + P0 + g * 8;
        const size_t vstep = dbg == 8 ? 16 * 32 : (size_t)16 * ctx_stride;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) vr[dt] = ldc(vsrc + dt * vstep);
    };
    if (wid < nit) { load_k(wid); load_v(wid); }      // cache blocks start streaming before anything else

    // rotate-half RoPE of a head's four 8-wide slices held by this lane: d = sp*64 + g*16 + hf*8 + e (its partner
    // d +- 64 is the same lane's other sp), rounded to bf16 like the stored q / k
    auto rope_head = [&](int col, u32x4_t (&out)[4]) {
        u32x4_t x[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x[s4] = slice8(col + (s4 >> 1) * 64 + g * 16 + (s4 & 1) * 8);
        if (!fuse_rope) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) out[s4] = x[s4];
            return;
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)p_new * (HD / 2) + g * 16 + hf * 8);
            const float4* sq = reinterpret_cast<const float4*>(sin_t + (size_t)p_new * (HD / 2) + g * 16 + hf * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sq[0], s1 = sq[1];
            const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t a = x[hf][e], bb = x[2 + hf][e];
                const float x1l = bflo(a), x1h = bfhi(a), x2l = bflo(bb), x2h = bfhi(bb);
                out[hf][e] = pack2bf(x1l * cs[2 * e] - x2l * sn[2 * e], x1h * cs[2 * e + 1] - x2h * sn[2 * e + 1]);
                out[2 + hf][e] = pack2bf(x2l * cs[2 * e] + x1l * sn[2 * e], x2h * cs[2 * e + 1] + x1h * sn[2 * e + 1]);
            }
        }
    };
    u32x4_t qf[4];                                    // B operand of S: q of head i (i < 4), zero columns otherwise
    if (i < GQ) rope_head((kvh * GQ + i) * HD, qf);
    else {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = zero4;
    }
    if (owner) {       // the newest row: append to the caches (K row-major, V down a column of V^T) and park it in LDS
        if (wid == 0 && i == 0) {
            u32x4_t kn[4];
            rope_head((nq + kvh) * HD, kn);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int d0 = (s4 >> 1) * 64 + g * 16 + (s4 & 1) * 8;
                *reinterpret_cast<u32x4_t*>(kb + (size_t)p_new * HD + d0) = kn[s4];
                *reinterpret_cast<u32x4_t*>(&s_new[d0]) = kn[s4];
            }
        }
        if (tid >= NW * 64 - 128) {
            const int d = tid - (NW * 64 - 128);
            const bf16_t x = elem((nq + nkv + kvh) * HD + d);
            vb[(size_t)d * ctx_stride + p_new] = x;
            s_new[HD + d] = x;
        }
        __syncthreads();                              // workgroup-uniform branch
    }

    float m = -1e30f, l = 0.f;                        // running max (per head = per column i) and this lane's partial sum
    f32x4_t acc[8];                                   // O^T tile dt: lane (i, g) reg r  <->  d = dt*16 + g*4 + r, head i
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // single register set, refilled as soon as the MFMAs that read it have issued: the next block's K streams in
    // during this block's softmax + PV, its V^T during the next block's QK (3 waves/SIMD cover the rest)
    // The loop is in two parts (round 6): a steady part whose every iteration requests the next block WITHOUT a condition, and the wave's last block, which
    // requests nothing.  With `if (it + NW < nit)` around the requests the wait-count pass saw an unknown number of loads in flight at the loop header and waited
    // for all of them (s_waitcnt vmcnt(0)) in front of the QK MFMAs — the V^T block just requested included; now it waits for the K block only (vmcnt(8)).
    auto block = [&](const int it, auto prefetch) {
        const int P0 = beg + it * 32;
        // (the newest position is the split's last one: it can only lie in a wave's LAST block, so the steady iterations carry no splice — its partial register
        // writes and LDS reads at a join were the other thing that made the wait-count pass give up and wait for everything)
        if (!decltype(prefetch)::value && fuse_rope && p_new >= P0 && p_new < P0 + 32) {          // wave-uniform: splice the newest k / v (parked in LDS)
            const int o = p_new - P0;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (prow + 4 * t == o) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        kr[t][s4] = *reinterpret_cast<const u32x4_t*>(&s_new[(s4 >> 1) * 64 + g * 16 + (s4 & 1) * 8]);
                }
            if (g == (o >> 3)) {
                const int wsel = (o & 7) >> 1, hi = o & 1;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    const uint32_t x = s_new[HD + dt * 16 + i];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t old = vr[dt][w];
                        const uint32_t ins = hi ? ((old & 0xffffu) | (x << 16)) : ((old & 0xffff0000u) | x);
                        vr[dt][w] = (w == wsel) ? ins : old;
                    }
                }
            }
        }
        f32x4_t S[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            S[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                S[t] = mfma16(__builtin_bit_cast(bf16x8_t, kr[t][s4]),
                                                               __builtin_bit_cast(bf16x8_t, qf[s4]), S[t]);
        }
        if constexpr (decltype(prefetch)::value) load_k(it + NW);
        // lane (head i, group g): S[t][r] is the score of position P0 + g*8 + t*4 + r
        float sv[8];
        float mx = -1e30f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = P0 + g * 8 + e < end;
            sv[e] = ok ? S[e >> 2][e & 3] * scale : -1e30f;
            mx = fmaxf(mx, sv[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float a = __expf(m - mn);
        m = mn;
        float p[8], ps = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = (P0 + g * 8 + e < end) ? __expf(sv[e] - mn) : 0.f;
            ps += p[e];
        }
        l = l * a + ps;
        const u32x4_t pf = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
        if constexpr (!decltype(prefetch)::value) {
            // only a wave's LAST block can reach past `end`: 8-position groups wholly past it count as zeros, as they did when their loads were predicated
            // (p = 0 there, but 0 x a non-finite leftover of an earlier sequence would not be 0)
            const bool vok = P0 + g * 8 < end;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) vr[dt] = vok ? vr[dt] : zero4;
        }
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            acc[dt] *= a;
            acc[dt] = mfma16(__builtin_bit_cast(bf16x8_t, vr[dt]),
                                                              __builtin_bit_cast(bf16x8_t, pf), acc[dt]);
        }
        if constexpr (decltype(prefetch)::value) load_v(it + NW);
    };
    // Everything requested so far (the first K / V^T block, the q slices) is waited for HERE, with a wait the compiler's wait-count pass can see (the builtin, not
    // inline asm).  The q fragments are first used inside the loop; left pending at its entry they are the YOUNGEST loads in flight there, the one static wait at
    // the loop header has to cover that edge too, and it came out as s_waitcnt vmcnt(0) on EVERY iteration — each QK product waited for the V^T block requested
    // a moment earlier (ISA read, round 6; the first block is needed in full before its products anyway).  (vmcnt(0); expcnt / lgkmcnt untouched.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    {
        int it = wid;
        for (; it + NW < nit; it += NW) block(it, std::true_type{});
        if (it < nit) block(it, std::false_type{});
    }
    // ---- the k-groups of a wave share m; sum their l; then merge the 4 waves through LDS ----
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (i < GQ) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            *reinterpret_cast<f32x4_t*>(&s_acc[wid][i][dt * 16 + g * 4]) = acc[dt];
        if (g == 0) { s_m[wid][i] = m; s_l[wid][i] = l; }
    }
    __syncthreads();
    if (dbg == 2) return;
    {
        const size_t base = (((size_t)b * nq + kvh * GQ) * nsplit + sp) * (HD + 2);
        // write-through (sc1) stores: visible at agent scope once vmcnt drains, no L2 write-back fence needed
        for (int x = tid; x < GQ * HD; x += NW * 64) {
            const int hq = x >> 7, d = x & 127;
            float M = s_m[0][hq];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, s_m[w][hq]);
            float o = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = __expf(s_m[w][hq] - M);
                o += f * s_acc[w][hq][d];
                L += f * s_l[w][hq];
            }
            if (nsplit == 1) {                        // nothing to merge across workgroups: finish here
                O[(size_t)b * ldo + (kvh * GQ + hq) * HD + d] = f2bf(o / L);
                continue;
            }
            __hip_atomic_store(&ws[base + (size_t)hq * nsplit * (HD + 2) + d], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(&ws[base + (size_t)hq * nsplit * (HD + 2) + HD], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ws[base + (size_t)hq * nsplit * (HD + 2) + HD + 1], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (dbg == 1 || nsplit == 1) return;
    // ---- publish + ticket; the last arriver of this (b, kv-head) merges the splits ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(&tickets[b * nkv + kvh], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsplit - 1));
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_last = last;
    }
    __syncthreads();
    if (!s_last || dbg == 3) return;
    for (int hq = wid; hq < GQ; hq += NW) {   // wave w merges q-head w (w + NW): lane owns d = 2*lane, 2*lane+1; split loads are independent -> issued in batches
        const float* w = ws + (((size_t)b * nq + kvh * GQ + hq) * nsplit) * (HD + 2);
        float M = -1e30f;
        for (int s2 = lane; s2 < nsplit; s2 += 64) M = fmaxf(M, w[s2 * (HD + 2) + HD]);
        M = wave_max(M);
        float num0 = 0.f, num1 = 0.f, den = 0.f;
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            float2 o[8];
            float mm[8], ll[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s2 = min(s0 + u, nsplit - 1);
                const float* ws2 = w + (size_t)s2 * (HD + 2);
                o[u] = *reinterpret_cast<const float2*>(ws2 + 2 * lane);
                mm[u] = ws2[HD];
                ll[u] = ws2[HD + 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (s0 + u < nsplit) {
                    const float f = __expf(mm[u] - M);
                    num0 += f * o[u].x; num1 += f * o[u].y; den += f * ll[u];
                }
            }
        }
        const float inv = 1.f / den;
        *reinterpret_cast<uint32_t*>(O + (size_t)b * ldo + (kvh * GQ + hq) * HD + 2 * lane) = pack2bf(num0 * inv, num1 * inv);
    }
    if (tid == 0) __hip_atomic_store(&tickets[b * nkv + kvh], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------
// Heads.  Wh rows follow the global vocabulary [text 0..V-1 | <sync> V | time | score], padded to 16.
__device__ __forceinline__ void head_bounds(int head, int V, int Tv, int Sv, int& lo, int& hi) {
    lo = head == 0 ? 0 : (head == 1 ? V + 1 : V + 1 + Tv);
    hi = head == 0 ? V + 1 : (head == 1 ? V + 1 + Tv : V + 1 + Tv + Sv);
}

__global__ __launch_bounds__(512) void head_logits_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ Wh,
                                                          int H, const int32_t* __restrict__ heads, int V, int Tv, int Sv,
                                                          float* __restrict__ part_val, int32_t* __restrict__ part_idx,
                                                          float* __restrict__ logits_out, int B, int ntiles) {
    __shared__ float red[8][2][256];
    __shared__ float fin[2][256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int NV = V + 1 + Tv + Sv;
    const int NB = B > 16 ? 2 : 1;
    bool any = false;
    for (int b = 0; b < B; ++b) {
        int lo, hi;
        head_bounds(heads[b], V, Tv, Sv, lo, hi);
        any |= (n0 < hi) && (n0 + 16 > lo);
    }
    if (!any) {
        if (tid < B) { part_val[(size_t)tid * ntiles + blockIdx.x] = -INFINITY; part_idx[(size_t)tid * ntiles + blockIdx.x] = n0; }
        if (logits_out) {
            for (int i = tid; i < B * 16; i += 512) {
                const int b = i >> 4, n = n0 + (i & 15);
                if (n < NV) logits_out[(size_t)b * NV + n] = -INFINITY;
            }
        }
        return;
    }
    const int U = H >> 6;
    const int u0 = (wid * U) >> 3, u1 = ((wid + 1) * U) >> 3;
    const bf16_t* wp = Wh + (size_t)(n0 + r) * H + g * 16;
    const bool xon0 = r < B, xon1 = r + 16 < B;
    const bf16_t* xp0 = X + (size_t)(xon0 ? r : 0) * ldx + g * 16;
    const bf16_t* xp1 = X + (size_t)(xon1 ? r + 16 : 0) * ldx + g * 16;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int u = u0; u < u1; u += 4) {
        Frag w[4][2], x0[4][2], x1[4][2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const bool ok = u + jj < u1;
            const int ko = (u + jj) * 64;
            w[jj][0].u = ok ? ldg_nt(wp + ko) : make_uint4(0, 0, 0, 0);
            w[jj][1].u = ok ? ldg_nt(wp + ko + 8) : make_uint4(0, 0, 0, 0);
            x0[jj][0].u = (ok && xon0) ? *reinterpret_cast<const uint4*>(xp0 + ko) : make_uint4(0, 0, 0, 0);
            x0[jj][1].u = (ok && xon0) ? *reinterpret_cast<const uint4*>(xp0 + ko + 8) : make_uint4(0, 0, 0, 0);
            x1[jj][0].u = (ok && xon1) ? *reinterpret_cast<const uint4*>(xp1 + ko) : make_uint4(0, 0, 0, 0);
            x1[jj][1].u = (ok && xon1) ? *reinterpret_cast<const uint4*>(xp1 + ko + 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc0 = mfma16(w[jj][0].v, x0[jj][0].v, acc0);
            acc0 = mfma16(w[jj][1].v, x0[jj][1].v, acc0);
            if (NB == 2) {
                acc1 = mfma16(w[jj][0].v, x1[jj][0].v, acc1);
                acc1 = mfma16(w[jj][1].v, x1[jj][1].v, acc1);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[wid][0][i * 64 + lane] = acc0[i]; red[wid][1][i * 64 + lane] = acc1[i]; }
    __syncthreads();
    if (tid < 256) {
        // tid = i*64 + l  ->  m = l&15, n_local = (l>>4)*4 + i ; store as fin[nb][m*16 + n_local]
        const int i = tid >> 6, l = tid & 63;
        for (int nb = 0; nb < NB; ++nb) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w][nb][tid];
            fin[nb][(l & 15) * 16 + (l >> 4) * 4 + i] = s;
        }
    }
    __syncthreads();
    if (tid < B * 16) {
        const int b = tid >> 4, nl = tid & 15, n = n0 + nl;
        int lo, hi;
        head_bounds(heads[b], V, Tv, Sv, lo, hi);
        const bool ok = n >= lo && n < hi;
        float v = ok ? fin[b >> 4][(b & 15) * 16 + nl] : -INFINITY;
        if (logits_out && n < NV) logits_out[(size_t)b * NV + n] = v;
        // arg-max over the 16 rows of this tile for sequence b (lowest index wins ties)
        int idx = n;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if (nl == 0) { part_val[(size_t)b * ntiles + blockIdx.x] = v; part_idx[(size_t)b * ntiles + blockIdx.x] = idx; }
    }
}

// one workgroup per sequence.  `step` is shared: every workgroup reads it first, then checks in; the last one to
// check in advances it (so no workgroup can observe the next step's value).
__global__ __launch_bounds__(256) void select_next_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
                                                          StepState st, const bf16_t* __restrict__ embed,
                                                          const bf16_t* __restrict__ time_tab, const bf16_t* __restrict__ score_tab,
                                                          const bf16_t* __restrict__ sync_row, bf16_t* __restrict__ xnext, int ldx,
                                                          int B, int H, int V, int Tv, int Sv, int ntiles, int advance) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_feed;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    const int step = __hip_atomic_load(st.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int max_new = st.params[0], eos = st.params[1], record_feed = st.params[2];
    __syncthreads();
    if (tid == 0) {
        const int arrived = __hip_atomic_fetch_add(st.step + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == B - 1) {
            __hip_atomic_store(st.step + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.step, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int t = tid; t < ntiles; t += 256) {
        const float ov = part_val[(size_t)b * ntiles + t];
        const int oi = part_idx[(size_t)b * ntiles + t];
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (lane == 0) { sv[wid] = v; si[wid] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > v || (sv[w] == v && si[w] < idx)) { v = sv[w]; idx = si[w]; }
        int tok = idx;
        if (advance && step < max_new) st.pos[b] += 1;     // (the host bounds the step count too: engine.hip trace_decode_steps)
        int feed = tok;
        if (step < max_new) {
            const int f = st.forced[(size_t)b * max_new + step];
            if (f >= 0) feed = f;
        }
        if (record_feed) tok = feed;          // host-driven sampling: the emitted token is the one fed back
        const bool was_done = st.done[b] != 0;
        if (!was_done && step < max_new) {
            st.out_ids[(size_t)b * max_new + step] = tok;
            st.out_len[b] = step + 1;
            if (eos >= 0 && tok == eos) st.done[b] = 1;
        }
        // head switch (trace_mistral.py:86-88): V -> time(1), V+1 -> score(2), V+Tv+1 -> text(0)
        int hd = st.heads[b];
        if (feed == V) hd = 1; else if (feed == V + 1) hd = 2; else if (feed == V + Tv + 1) hd = 0;
        st.heads[b] = hd;
        s_feed = feed;
    }
    __syncthreads();
    const int feed = s_feed;
    const bf16_t* src;
    if (feed == V) src = sync_row;
    else if (feed > V && feed < V + 1 + Tv) src = time_tab + (size_t)(feed - V - 1) * H;
    else if (feed >= V + 1 + Tv) src = score_tab + (size_t)(feed - V - 1 - Tv) * H;
    else src = embed + (size_t)(feed % V) * H;
    for (int c = tid; c < (H >> 3); c += 256)
        *reinterpret_cast<uint4*>(xnext + (size_t)b * ldx + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
}
}  // namespace

int g_skinny_debug = 0;   // microbenchmark-only: 3 = stop before the epilogue, 4 = stop after the partial stores (ticket path)
// Partition for skinny_lds (see the kernel header): KS k-chunks x row-groups of T tiles, WPT waves per tile.
struct SkinnyPlan { int KS, chunk_units, T, WPT, ntiles, grid, threads; };
static int skinny_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
static int skinny_nt(int N, int epi) { return epi == EPI_SWIGLU || (epi == EPI_PARTIAL && N >= 16384 && N % 32 == 0) ? 2 : 1; }
static int skinny_nb(int B) { return B > 32 ? 4 : B > 16 ? 2 : 1; }       // 16-row activation groups
static SkinnyPlan skinny_plan(int N, int K, int epi, int B) {
    SkinnyPlan p{};
    const int NT = skinny_nt(N, epi), NB = skinny_nb(B);
    const int U = K / 64;
    static const int cap1 = getenv("TRACE_SK_CAP") ? atoi(getenv("TRACE_SK_CAP")) : 64;    // tuning knob (microbenchmarks)
    const int cap = NB == 1 ? cap1 : 64 / NB;             // units of X that fit 128 KB of LDS
    const int ks_min = (U + cap - 1) / cap;
    const int ks_max = epi == EPI_PARTIAL ? std::min(U, ks_min + 4) : ks_min;    // extra chunks are free only without the ticket merge
    p.ntiles = N / (16 * NT);
    const int ncu = skinny_num_cus();
    long best = -1;
    for (int ks = ks_min; ks <= ks_max; ++ks) {
        const int chunk = (U + ks - 1) / ks;
        for (int T = 1; T <= 16; ++T) {
            const int grid = ks * ((p.ntiles + T - 1) / T);
            const long rounds = (grid + ncu - 1) / ncu;
            const long cost = rounds * T * chunk * 64 + rounds * 8 + ks;   // per-CU weight stream; ties -> fewer rounds, fewer chunks
            if (best < 0 || cost < best) { best = cost; p.KS = ks; p.chunk_units = chunk; p.T = T; p.grid = grid; }
        }
    }
    p.WPT = 1;
    // T*WPT <= 8 waves; the LDS reduction scratch (T*WPT*NT*NB KB) must fit beside the 128 KB of parked activations
    while (p.WPT * 2 * p.T <= 8 && p.WPT * 2 * p.T * NT * NB <= 28 && p.WPT * 2 <= p.chunk_units) p.WPT *= 2;
    p.threads = std::min(p.T, 8) * p.WPT * 64;          // T > 8: every wave takes two tasks in turn
    return p;
}
static size_t skinny_plan_ws(const SkinnyPlan& p, int N, int epi, int B) {
    if (epi == EPI_PARTIAL) return (size_t)p.KS * SK_ROWS * N;
    const int NT = skinny_nt(N, epi), NB = skinny_nb(B);
    return p.KS > 1 ? (size_t)p.ntiles * p.KS * NT * NB * 256 : 0;
}
size_t skinny_ws_floats(int N, int K, int epi) {
    size_t f = 0;
    for (int B : {1, 32, 64}) f = std::max(f, skinny_plan_ws(skinny_plan(N, K, epi, B), N, epi, B));
    return f;
}
int skinny_ks(int N, int K, int epi, int B) { return skinny_plan(N, K, epi, B).KS; }

template <int EPI, int NB, int NT, int PRO = 0>
static int skinny_lds_launch(const SkinnyPlan& p, const bf16_t* X, int ldx, const bf16_t* W, int ldw, bf16_t* out, int ldo,
                             const bf16_t* R, int ldr, int B, int K, float* ws, unsigned int* tickets, int tiled, hipStream_t s,
                             const SkinnyPro& pro = SkinnyPro{}) {
    const size_t lds = (size_t)p.chunk_units * 2 * NB * 1024 + (p.WPT > 1 ? (size_t)p.T * p.WPT * NT * NB * 1024 : 0);
    static LdsGrantSized grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(skinny_lds_kernel<EPI, NB, NT, PRO>), lds)) return TRACE_ERR_HIP;
    hipLaunchKernelGGL((skinny_lds_kernel<EPI, NB, NT, PRO>), dim3(p.grid), dim3(p.threads), lds, s, X, ldx, W, ldw, out, ldo, R, ldr, B, K,
                       p.chunk_units, p.KS, p.T, p.WPT, p.ntiles, ws, tickets, tiled, g_skinny_debug, pro);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

bool skinny_fused_norm_ok(int N, int K, int B) {
    if (B < 1 || B > SKINNY_PRO_ROWS || K % 64 || K > 16384 || N % 16) return false;
    return (K >> 3) <= 2 * skinny_plan(N, K, EPI_PARTIAL, B).threads;
}
// The fused form (see SkinnyPro): out-partials[ks][SK_ROWS][N] = RMSNorm(sum_k part_in + R; w) . Wtiled^T, new residual rows to xout.  B <= SKINNY_PRO_ROWS,
// K <= 16384, tiled weights, part_in / ws and R / xout must not alias.
int launch_skinny_gemm_fused_norm(const float* part_in, int ks_in, const bf16_t* R, int ldr, bf16_t* xout, int ldx, const bf16_t* w, float eps,
                                  const bf16_t* Wtiled, int B, int N, int K, float* ws, size_t ws_floats, hipStream_t s) {
    if (B < 1 || B > SKINNY_PRO_ROWS || K % 64 || K > 16384 || N % 16 || ks_in < 0 || (ks_in > 0 && !part_in) || !R || !xout || R == xout || !w) return TRACE_ERR_ARG;
    if (part_in == ws || (ldr % 8) || (ldx % 8)) return TRACE_ERR_ARG;
    const SkinnyPlan p = skinny_plan(N, K, EPI_PARTIAL, B);
    if (!ws || ws_floats < skinny_plan_ws(p, N, EPI_PARTIAL, B)) return TRACE_ERR_ARG;
    if ((K >> 3) > 2 * p.threads) return TRACE_ERR_STATE;                     // MAXG groups per thread: the caller keeps the unfused pair for this shape
    const SkinnyPro pro{part_in, ks_in, R, ldr, xout, ldx, w, eps};
    return skinny_nt(N, EPI_PARTIAL) == 2 ? skinny_lds_launch<EPI_PARTIAL, 1, 2, 1>(p, nullptr, K, Wtiled, K, nullptr, N, nullptr, 0, B, K, ws, nullptr, 1, s, pro)
                                           : skinny_lds_launch<EPI_PARTIAL, 1, 1, 1>(p, nullptr, K, Wtiled, K, nullptr, N, nullptr, 0, B, K, ws, nullptr, 1, s, pro);
}

// The down GEMV with the SwiGLU prologue (SkinnyPro, PRO == 2): out-partials[ks][SK_ROWS][N] = (silu(sum_k gate) * sum_k up) . Wtiled^T from the
// gate|up GEMV's partial rows part_gu [ks_gu][SK_ROWS][2 K].  B <= SKINNY_PRO_ROWS, tiled weights, part_gu and ws must not alias.
int launch_skinny_gemm_fused_swiglu(const float* part_gu, int ks_gu, const bf16_t* Wtiled, int B, int N, int K, float* ws, size_t ws_floats, hipStream_t s) {
    if (B < 1 || B > SKINNY_PRO_ROWS || K % 64 || N % 16 || ks_gu < 1 || !part_gu || part_gu == ws) return TRACE_ERR_ARG;
    const SkinnyPlan p = skinny_plan(N, K, EPI_PARTIAL, B);
    if (!ws || ws_floats < skinny_plan_ws(p, N, EPI_PARTIAL, B)) return TRACE_ERR_ARG;
    const SkinnyPro pro{part_gu, ks_gu, nullptr, 0, nullptr, 0, nullptr, 0.f};
    return skinny_nt(N, EPI_PARTIAL) == 2 ? skinny_lds_launch<EPI_PARTIAL, 1, 2, 2>(p, nullptr, K, Wtiled, K, nullptr, N, nullptr, 0, B, K, ws, nullptr, 1, s, pro)
                                           : skinny_lds_launch<EPI_PARTIAL, 1, 1, 2>(p, nullptr, K, Wtiled, K, nullptr, N, nullptr, 0, B, K, ws, nullptr, 1, s, pro);
}

// EPI_PARTIAL: `out` is unused, ldo = N, the fp32 partial rows [KS = skinny_ks()][SK_ROWS][N] land in ws.
int launch_skinny_gemm(const bf16_t* X, int ldx, const bf16_t* W, int ldw, bf16_t* out, int ldo, const bf16_t* R, int ldr,
                       int B, int N, int K, int epi, int tiled, float* ws, size_t ws_floats, unsigned int* tickets, int ntickets,
                       hipStream_t s) {
    if (B < 1 || B > SKINNY_ROWS || K % 64 || (ldx % 8) || (ldw % 8) || (ldo % 4)) return TRACE_ERR_ARG;
    if (epi != EPI_NONE && epi != EPI_RESIDUAL && epi != EPI_SWIGLU && epi != EPI_PARTIAL) return TRACE_ERR_ARG;
    if (N % (epi == EPI_SWIGLU ? 32 : 16) || (epi == EPI_RESIDUAL && (!R || ldr % 4))) return TRACE_ERR_ARG;
    const SkinnyPlan p = skinny_plan(N, K, epi, B);
    const size_t need = skinny_plan_ws(p, N, epi, B);
    if (need && (!ws || ws_floats < need)) return TRACE_ERR_ARG;
    if (epi != EPI_PARTIAL && p.KS > 1 && (!tickets || ntickets < p.ntiles)) return TRACE_ERR_ARG;
    if (epi == EPI_PARTIAL) ldo = N;
#define SL(EPI_, NT_) (B <= 16 ? skinny_lds_launch<EPI_, 1, NT_>(p, X, ldx, W, ldw, out, ldo, R, ldr, B, K, ws, tickets, tiled, s) \
                     : B <= 32 ? skinny_lds_launch<EPI_, 2, NT_>(p, X, ldx, W, ldw, out, ldo, R, ldr, B, K, ws, tickets, tiled, s) \
                               : skinny_lds_launch<EPI_, 4, NT_>(p, X, ldx, W, ldw, out, ldo, R, ldr, B, K, ws, tickets, tiled, s))
    switch (epi) {
        case EPI_NONE: return SL(EPI_NONE, 1);
        case EPI_RESIDUAL: return SL(EPI_RESIDUAL, 1);
        case EPI_PARTIAL: return skinny_nt(N, epi) == 2 ? SL(EPI_PARTIAL, 2) : SL(EPI_PARTIAL, 1);
        default: return SL(EPI_SWIGLU, 2);
    }
#undef SL
}

int launch_swiglu_combine(const float* part, int KS, int N2, bf16_t* out, int ldo, int B, hipStream_t s) {
    if (B < 1 || B > SK_ROWS || KS < 1 || N2 % 32 || ldo % 4) return TRACE_ERR_ARG;
    const int total = B * (N2 / 8);
    hipLaunchKernelGGL(swiglu_combine_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, KS, N2, out, ldo, B);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_tile_pack(const bf16_t* src, int ldw, bf16_t* dst, int N, int K, hipStream_t s) {
    if (N % 16 || K % 64 || ldw % 8) return TRACE_ERR_ARG;
    const long total = (long)N * (K / 8);
    const int grid = (int)std::min<long>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(tile_pack_kernel, dim3(grid), dim3(256), 0, s, src, ldw, dst, N, K);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_add_rmsnorm(const float* part, int KS, const bf16_t* R, int ldr, bf16_t* xout, int ldx, const bf16_t* w, bf16_t* y,
                       int ldy, int B, int N, float eps, hipStream_t s, uint8_t* y8, float* sy) {
    if (B < 1 || B > SK_ROWS || KS < 1 || N % 4 || N > 4096 || (ldr % 4) || (ldx % 4) || (ldy % 4)) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(B), dim3(1024), 0, s, part, KS, R, ldr, xout, ldx, w, y, ldy, N, eps, y8, sy);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

// The qkv GEMV's fp32 partial rows -> bf16 q (RoPE applied) in `qout`, the new k row (RoPE applied) and v column appended to the caches:
// what attn_decode_kernel's fused prologue does, as a kernel of its own (8 lanes per (sequence, head): lane c holds d = 8c..8c+7 and its
// rotate-half partner d + 64).  Same sums (chunk order), same roundings as the fused path.
__global__ __launch_bounds__(256) void qkv_finish_kernel(const float* __restrict__ part, int ks, int ldq, bf16_t* __restrict__ qout,
                                                         bf16_t* __restrict__ kcache, bf16_t* __restrict__ vtcache, long slot_stride,
                                                         long kv_head_stride, int ctx_stride, const int32_t* __restrict__ slots,
                                                         const int32_t* __restrict__ pos, int B, int nq, int nkv,
                                                         const float* __restrict__ cos_t, const float* __restrict__ sin_t) {
    constexpr int HD = 128, HALF = 64;
    const int nh = nq + 2 * nkv;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * nh * 8) return;
    const int c = idx & 7, hh = (idx >> 3) % nh, b = (idx >> 3) / nh;
    const int p_new = pos[b];
    const float* pp = part + (size_t)b * ldq + (size_t)hh * HD + c * 8;
    f32x4_t x[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // d..d+3, d+4..d+7, d+64.., d+68..
    for (int k = 0; k < ks; ++k) {
        const float* pk = pp + (size_t)k * SK_ROWS * ldq;
        x[0] += *reinterpret_cast<const f32x4_t*>(pk);
        x[1] += *reinterpret_cast<const f32x4_t*>(pk + 4);
        x[2] += *reinterpret_cast<const f32x4_t*>(pk + HALF);
        x[3] += *reinterpret_cast<const f32x4_t*>(pk + HALF + 4);
    }
    uint32_t a[4] = {pack2bf(x[0][0], x[0][1]), pack2bf(x[0][2], x[0][3]), pack2bf(x[1][0], x[1][1]), pack2bf(x[1][2], x[1][3])};
    uint32_t bb[4] = {pack2bf(x[2][0], x[2][1]), pack2bf(x[2][2], x[2][3]), pack2bf(x[3][0], x[3][1]), pack2bf(x[3][2], x[3][3])};
    if (hh >= nq + nkv) {                                // v: down a column of V^T
        bf16_t* vb = vtcache + (size_t)slots[b] * slot_stride + (size_t)(hh - nq - nkv) * kv_head_stride + p_new;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vb[(size_t)(c * 8 + 2 * e) * ctx_stride] = (bf16_t)(a[e] & 0xffffu);
            vb[(size_t)(c * 8 + 2 * e + 1) * ctx_stride] = (bf16_t)(a[e] >> 16);
            vb[(size_t)(HALF + c * 8 + 2 * e) * ctx_stride] = (bf16_t)(bb[e] & 0xffffu);
            vb[(size_t)(HALF + c * 8 + 2 * e + 1) * ctx_stride] = (bf16_t)(bb[e] >> 16);
        }
        return;
    }
    const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)p_new * HALF + c * 8);
    const float4* sp = reinterpret_cast<const float4*>(sin_t + (size_t)p_new * HALF + c * 8);
    const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint32_t o1[4], o2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x1l = bflo(a[e]), x1h = bfhi(a[e]), x2l = bflo(bb[e]), x2h = bfhi(bb[e]);
        o1[e] = pack2bf(x1l * cs[2 * e] - x2l * sn[2 * e], x1h * cs[2 * e + 1] - x2h * sn[2 * e + 1]);
        o2[e] = pack2bf(x2l * cs[2 * e] + x1l * sn[2 * e], x2h * cs[2 * e + 1] + x1h * sn[2 * e + 1]);
    }
    bf16_t* dst = hh < nq ? qout + (size_t)b * ldq + (size_t)hh * HD
                          : kcache + (size_t)slots[b] * slot_stride + (size_t)(hh - nq) * kv_head_stride + (size_t)p_new * HD;
    *reinterpret_cast<uint4*>(dst + c * 8) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    *reinterpret_cast<uint4*>(dst + HALF + c * 8) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}
int launch_qkv_finish(const float* part, int ks, int ldq, bf16_t* qout, bf16_t* kcache, bf16_t* vtcache, long slot_stride, long kv_head_stride,
                      int ctx_stride, const int32_t* slots, const int32_t* pos, int B, int nq, int nkv, const float* cos_t, const float* sin_t,
                      hipStream_t s) {
    if (B < 1 || ks < 1 || (ldq % 8)) return TRACE_ERR_ARG;
    const int total = B * (nq + 2 * nkv) * 8;
    hipLaunchKernelGGL(qkv_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, ks, ldq, qout, kcache, vtcache, slot_stride,
                       kv_head_stride, ctx_stride, slots, pos, B, nq, nkv, cos_t, sin_t);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int g_attn_debug = 0;   // microbenchmark-only phase cut-offs (0 = full kernel)
int g_attn_decode_w3 = -1;  // waves per decode-attention workgroup (A/B: trace_op_set_gemm_variant(760 + x)): 0 / 2 (-1) = 4 (ships), 1 = 3, 3 = 6, 4 = 8
int g_attn_decode_lds_pad = 0;   // KB of unused dynamic LDS per decode-attention workgroup (A/B: trace_op_set_gemm_variant(780 + KB / 8))
int g_attn_decode_nt = 0;   // non-temporal cache loads in the decode attention (A/B: trace_op_set_gemm_variant(770 + x))
int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* kcache, bf16_t* vtcache, long slot_stride, long kv_head_stride,
                       int ctx_stride, const int32_t* slots, const int32_t* pos, bf16_t* O, int ldo, float* ws, unsigned int* tickets, int B,
                       int nq, int nkv, int hd, int nsplit, float scale, int fuse_rope, const float* cos_t, const float* sin_t,
                       const float* qpart, int qks, hipStream_t s) {
    if (hd != 128 || nq != 4 * nkv || nsplit < 1 || B < 1 || !tickets || ctx_stride % 32) return TRACE_ERR_ARG;
    // 3-wave workgroups when the 4-wave form would not be resident in one round (3 x #CUs slots) but the 3-wave form is (4 x #CUs)
    const long wgs = (long)nsplit * nkv * B;
    const int ncu = skinny_num_cus();
    const int nw = g_attn_decode_w3 < 0 ? 4 : g_attn_decode_w3 == 1 ? 3 : g_attn_decode_w3 == 3 ? 6 : g_attn_decode_w3 == 4 ? 8 : 4;
    (void)wgs; (void)ncu;
    const size_t pad = (size_t)g_attn_decode_lds_pad * 1024;       // A/B: dynamic LDS nobody uses, caps the workgroups per CU
#define ATTN_DEC(NW_, NT_) do { static LdsGrantSized grant_; if (pad && !grant_dynamic_lds(grant_, reinterpret_cast<const void*>(attn_decode_kernel<NW_, NT_>), pad)) return TRACE_ERR_HIP; \
    hipLaunchKernelGGL((attn_decode_kernel<NW_, NT_>), dim3(nsplit, nkv, B), dim3(NW_ * 64), pad, s, qkv, ldq, kcache, vtcache, slot_stride, \
                       kv_head_stride, ctx_stride, slots, pos, ws, tickets, O, ldo, nq, nkv, nsplit, scale, fuse_rope, cos_t, sin_t, qpart, qks, g_attn_debug); } while (0)
    if (nw == 3) { if (g_attn_decode_nt) ATTN_DEC(3, true); else ATTN_DEC(3, false); }
    else if (nw == 6) ATTN_DEC(6, false);
    else if (nw == 8) ATTN_DEC(8, false);
    else { if (g_attn_decode_nt) ATTN_DEC(4, true); else ATTN_DEC(4, false); }
#undef ATTN_DEC
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_head_logits(const bf16_t* X, int ldx, const bf16_t* Wh, int H, const int32_t* heads, int V, int Tv, int Sv,
                       float* part_val, int32_t* part_idx, float* logits_out, int B, hipStream_t s) {
    if (B < 1 || B > SK_ROWS || H % 64) return TRACE_ERR_ARG;
    const int ntiles = (V + 1 + Tv + Sv + 15) / 16, NV = V + 1 + Tv + Sv;
    for (int b0 = 0; b0 < B; b0 += 32)        // the kernel holds 32 rows; a larger batch takes a second pass over the active tiles
        hipLaunchKernelGGL(head_logits_kernel, dim3(ntiles), dim3(512), 0, s, X + (size_t)b0 * ldx, ldx, Wh, H, heads + b0, V, Tv, Sv,
                           part_val + (size_t)b0 * ntiles, part_idx + (size_t)b0 * ntiles,
                           logits_out ? logits_out + (size_t)b0 * NV : nullptr, std::min(32, B - b0), ntiles);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_select_next(const float* part_val, const int32_t* part_idx, const StepState& st, const bf16_t* embed,
                       const bf16_t* time_tab, const bf16_t* score_tab, const bf16_t* sync_row, bf16_t* xnext, int ldx, int B,
                       int H, int V, int Tv, int Sv, int advance, hipStream_t s) {
    if (B < 1 || B > SK_ROWS || H % 8) return TRACE_ERR_ARG;
    const int ntiles = (V + 1 + Tv + Sv + 15) / 16;
    hipLaunchKernelGGL(select_next_kernel, dim3(B), dim3(256), 0, s, part_val, part_idx, st, embed, time_tab, score_tab,
                       sync_row, xnext, ldx, B, H, V, Tv, Sv, ntiles, advance);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
