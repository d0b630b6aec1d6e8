// The batch-1 decode step as ONE persistent launch (round 6).
//
// The reference's drivers decode one video at a time (trace/eval/evaluate.py:298-357, scripts/inference/inference.py:32-128): every token is 32 layers x
// [qkv GEMV -> attention -> o GEMV -> gate|up GEMV -> down GEMV] = 160 dependent kernels of 10-45 us that each stream a slice of the 14 GB of weights.
// As separate launches (decode_step_fused, engine.hip) every kernel pays its own fill and drain: the next kernel's first weight bytes are requested
// only after the previous kernel's last workgroup has retired.  Here one workgroup per CU walks all 160 phases; between two phases sits a grid barrier,
// and a workgroup requests the first weight tiles of phase n + 1 (registers: two load batches per wave, 64-128 KB per CU) BEFORE it waits at the barrier
// that ends phase n, so HBM keeps streaming while the arrivals are counted.  Weights are read-only: only the small activations (fp32 partial rows, the
// residual row, the attention output) cross the barrier, behind an agent-scope release by the producer workgroup and an acquire by the consumer.
//
// The phases are the arithmetic of the shipped kernels, statement for statement: skinny_lds_kernel<EPI_PARTIAL, 1, NT, PRO> (decode.hip) with the same
// SkinnyPlan partition (k-chunks, tasks, waves per task) and attn_decode_kernel<4> with the fused RoPE / append prologue and the ticket merge of the
// context splits — so the ids AND the logits are bit-identical to the launch-per-kernel path (tests/test_gpu_parity.py::test_batch1_persistent_step_*).
//
// Grid barrier (MI355X_MICROARCH.md "barrier-xcd", placement-independent form): workgroup w belongs to group w % 8 (= its XCD when the dispatcher deals
// workgroups round-robin; correctness does not depend on that), arrives with a fire-and-forget atomic on its group's counter; workgroups 0..7 collect
// their group, meet on a top counter and publish their group's generation word, which the other members poll (one lane, relaxed agent-scope loads,
// s_sleep).  Counters are monotonic within a launch (zeroed by a memset node in front of it); every spin is bounded — a timeout raises an error word
// and the launch drains without hanging the device.
#include "common.h"
#include "gridbar.h"
#include "kernels.h"

namespace {
using namespace gridbar;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 ldg_nt(const bf16_t* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
union Frag { uint4 u; bf16x8_t v; };
typedef DecodeB1Plan B1Plan;
typedef DecodeB1Layer B1Layer;
typedef DecodeB1Args B1Args;

struct Pro {                                   // skinny_lds_kernel's SkinnyPro
    const float* part; int ks;
    const bf16_t* R; bf16_t* xout;
    const bf16_t* w; float eps;
};

// One GEMV phase for ONE activation row: skinny_lds_kernel<EPI_PARTIAL, NB = 1, NT, PRO> with tiled weights, B = 1.  Waves beyond the plan's thread count
// and workgroups beyond its grid only keep the barriers company.
template <int NT, int PRO>
__device__ __forceinline__ void gemv_phase(unsigned char* smem, const B1Plan& P, const bf16_t* X, const bf16_t* W, int N, int K, float* ws, const Pro& pro,
                                           GridBar& gb, bool wait_first, int prefetch, int* s_flag, float* s_ss) {
    constexpr int UN = (NT == 2) ? 2 : 4;
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);                                          // [unit][half][lane]
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + (size_t)P.chunk_units * 2 * 1024);       // [task][wsub][NT][lane]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwaves = P.threads >> 6, nthr = P.threads;
    const int r = lane & 15, g = lane >> 4;
    const bool real = (int)blockIdx.x < P.grid && wid < nwaves && !gb.dead;
    const int ks = blockIdx.x % P.KS, rg = blockIdx.x / P.KS;
    const int U = K >> 6;
    const int u_beg = ks * P.chunk_units, nu = min(U - u_beg, P.chunk_units);
    const int WPT = P.WPT;
    const int team = wid / WPT, wsub = wid - team * WPT, nteams = nwaves / WPT;
    const int ua = (wsub * nu) / WPT, ub = ((wsub + 1) * nu) / WPT;
    const bf16_t* wp[NT];
    Frag wa[UN][NT][2], wb[UN][NT][2];
    auto loadw = [&](Frag (&wf)[UN][NT][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const bool ok = u + j < ub;
            const int ko = (u + j) * 1024;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wf[j][t][0].u = ok ? ldg_nt(wp[t] + ko) : make_uint4(0, 0, 0, 0);
                wf[j][t][1].u = ok ? ldg_nt(wp[t] + ko + 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    int task = team;
    int tile = rg * P.T + task;
    bool active = real && task < P.T && tile < P.ntiles;
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + ((size_t)(active ? tile * NT + t : 0) * U + u_beg) * 1024 + lane * 16;
    bool work = active && ua < ub;
    bool wb_ready = false;
    // the weight stream starts before the barrier that releases this phase's activations (weights depend on nothing)
    const int depth = prefetch & 3;
    // (opt bit 1: the polling wave keeps its vector-memory queue empty — a poll's reply returns behind every load the wave issued before it)
    const bool early = !wait_first || (depth > 0 && !((gb.opt & 1) && wid == 0));
    if (work && early) {
        loadw(wa, ua);
        if (depth > 1 && ua + UN < ub) { loadw(wb, ua + UN); wb_ready = true; }
    }
    if (wait_first) bar_wait(gb, s_flag);
    if (gb.dead) return;
    if (work && !early) loadw(wa, ua);

    // ---- park the activation row's K-chunk in LDS in fragment order: combo c = unit*2 + half, lane (r, g) holds X[(u_beg + unit)*64 + g*16 + half*8 .. +8] ----
    if constexpr (PRO == 1) {
        // sum of the previous GEMV's partial rows + residual -> new residual, RMSNorm (add_rmsnorm_kernel's arithmetic; every workgroup sums the whole row)
        const int ngrp = K >> 3;
        constexpr int MAXG = 2;
        uint4 xv[MAXG];
        float ss = 0.f;
        if (real) {
#pragma unroll
            for (int q = 0; q < MAXG; ++q) {
                const int e8 = tid + q * nthr;
                xv[q] = make_uint4(0u, 0u, 0u, 0u);
                if (e8 < ngrp) {
                    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                    const float* pp = pro.part + e8 * 8;
                    for (int k2 = 0; k2 < pro.ks; ++k2) {      // chunk order
                        a0 += *reinterpret_cast<const f32x4_t*>(pp + (size_t)k2 * SK_ROWS * K);
                        a1 += *reinterpret_cast<const f32x4_t*>(pp + (size_t)k2 * SK_ROWS * K + 4);
                    }
                    const uint4 rr = *reinterpret_cast<const uint4*>(pro.R + e8 * 8);
                    uint4 xo = rr;
                    if (pro.ks > 0) {
                        xo.x = pack2bf(bf2f(f2bf(a0[0])) + bflo(rr.x), bf2f(f2bf(a0[1])) + bfhi(rr.x));
                        xo.y = pack2bf(bf2f(f2bf(a0[2])) + bflo(rr.y), bf2f(f2bf(a0[3])) + bfhi(rr.y));
                        xo.z = pack2bf(bf2f(f2bf(a1[0])) + bflo(rr.z), bf2f(f2bf(a1[1])) + bfhi(rr.z));
                        xo.w = pack2bf(bf2f(f2bf(a1[2])) + bflo(rr.w), bf2f(f2bf(a1[3])) + bfhi(rr.w));
                    }
                    xv[q] = xo;
                    const uint32_t u[4] = {xo.x, xo.y, xo.z, xo.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) ss = fmaf(bflo(u[e]), bflo(u[e]), fmaf(bfhi(u[e]), bfhi(u[e]), ss));
                    const int unit = e8 >> 3;
                    if (rg == 0 && unit >= u_beg && unit < u_beg + nu) st_wt16(pro.xout, (size_t)e8 * 16, u32x4_t{xo.x, xo.y, xo.z, xo.w});
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) s_ss[wid] = ss;
        }
        __syncthreads();
        if (real) {
            float tot = 0.f;
            for (int w2 = 0; w2 < nwaves; ++w2) tot += s_ss[w2];
            const float rstd = rsqrtf(tot / (float)K + pro.eps);
#pragma unroll
            for (int q = 0; q < MAXG; ++q) {
                const int e8 = tid + q * nthr, unit = e8 >> 3;
                if (e8 < ngrp && unit >= u_beg && unit < u_beg + nu) {
                    const uint4 wv = *reinterpret_cast<const uint4*>(pro.w + e8 * 8);
                    const uint32_t xu[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w}, wu[4] = {wv.x, wv.y, wv.z, wv.w};
                    u32x4_t y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = pack2bf(bflo(xu[e]) * rstd * bflo(wu[e]), bfhi(xu[e]) * rstd * bfhi(wu[e]));
                    const int g2 = (e8 & 7) >> 1, half = e8 & 1;
                    xs[((unit - u_beg) * 2 + half) * 64 + g2 * 16 + 0] = y;          // row b = 0
                }
            }
        }
        __syncthreads();
    } else if constexpr (PRO == 2) {
        // SwiGLU over the gate|up GEMV's partial rows [pro.ks][SK_ROWS][2 K] (16-row interleaved) for THIS K-chunk (swiglu_combine_kernel's arithmetic)
        if (real) {
            const int ngrp = nu * 8, N2 = 2 * K;
            for (int idx = tid; idx < ngrp; idx += nthr) {
                const int e8l = idx, e8 = u_beg * 8 + e8l;
                const float* base = pro.part + (size_t)(e8 >> 1) * 32 + (e8 & 1) * 8;
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
                const int unit = e8l >> 3, g2 = (e8l & 7) >> 1, half = e8l & 1;
                xs[(unit * 2 + half) * 64 + g2 * 16 + 0] = y;
            }
        }
        __syncthreads();
    } else {
        // bf16 activations parked by LDS-DMA (every r-lane reads row 0: their accumulator rows are never stored)
        if (real) {
            const int combos = nu * 2;
            for (int c = wid; c < combos; c += nwaves) {
                const int h = c & 1, u = c >> 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)(u_beg + u) * 64 + g * 16 + h * 8),
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)c * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    f32x4_t acc[NT];
    auto mma = [&](Frag (&wf)[UN][NT][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            if (u + j < ub) {
                const bf16x8_t x0 = __builtin_bit_cast(bf16x8_t, xs[((u + j) * 2 + 0) * 64 + lane]);
                const bf16x8_t x1 = __builtin_bit_cast(bf16x8_t, xs[((u + j) * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = mfma16(wf[j][t][0].v, x0, acc[t]);
                    acc[t] = mfma16(wf[j][t][1].v, x1, acc[t]);
                }
            }
        }
    };
    bool first = true;
    for (;;) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (work) {
            for (int u = ua; u < ub; u += 2 * UN) {
                if (u + UN < ub && !(wb_ready && u == ua)) loadw(wb, u + UN);
                mma(wa, u);
                if (u + UN < ub) {
                    if (u + 2 * UN < ub) loadw(wa, u + 2 * UN);
                    mma(wb, u + UN);
                }
            }
        }
        if (first && WPT > 1) {          // WPT > 1: nteams == T, one task per team — the barrier is workgroup-uniform
            if (active) {
#pragma unroll
                for (int t = 0; t < NT; ++t) red[((team * WPT + wsub) * NT + t) * 64 + lane] = acc[t];
            }
            __syncthreads();
            if (active && wsub == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4_t sacc = red[((team * WPT) * NT + t) * 64 + lane];
                    for (int w = 1; w < WPT; ++w) sacc += red[((team * WPT + w) * NT + t) * 64 + lane];
                    acc[t] = sacc;
                }
            }
        }
        if (active && wsub == 0 && r == 0) {          // row m = r = 0 of the fragment: out[n0 + t*16 + g*4 .. +4]
            const size_t po = ((size_t)ks * SK_ROWS * N + tile * 16 * NT) * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t) st_wt16(ws, po + (size_t)(t * 16 + g * 4) * 4, __builtin_bit_cast(u32x4_t, acc[t]));
        }
        first = false;
        task += nteams;
        if (!real || task >= P.T) break;
        tile = rg * P.T + task;
        active = tile < P.ntiles;
#pragma unroll
        for (int t = 0; t < NT; ++t) wp[t] = W + ((size_t)(active ? tile * NT + t : 0) * U + u_beg) * 1024 + lane * 16;
        work = active && ua < ub;
        wb_ready = false;
        if (work) loadw(wa, ua);
    }
}

// The attention phase: attn_decode_kernel<4> (decode.hip) with fuse_rope = 1 and the qkv row as fp32 partial rows, for sequence 0; workgroup vb < nsplit * nkv
// takes (split, kv head) = (vb % nsplit, vb / nsplit); waves 4.. and the other workgroups only pass the barriers.
__device__ __attribute__((noinline)) void attn_phase(unsigned char* smem, const B1Args& A, const B1Layer& Lw, const float* qpart, int qks, GridBar& gb, int* s_flag) {
    constexpr int HD = 128, GQ = 4, NW = 4;
    float (*s_acc)[GQ][HD] = reinterpret_cast<float (*)[GQ][HD]>(smem);                              // [NW][GQ][HD]
    float (*s_m)[GQ] = reinterpret_cast<float (*)[GQ]>(smem + NW * GQ * HD * 4);
    float (*s_l)[GQ] = reinterpret_cast<float (*)[GQ]>(smem + NW * GQ * HD * 4 + NW * GQ * 4);
    bf16_t* s_new = reinterpret_cast<bf16_t*>(smem + NW * GQ * HD * 4 + 2 * NW * GQ * 4);          // [2 HD]
    int* s_last = reinterpret_cast<int*>(smem + NW * GQ * HD * 4 + 2 * NW * GQ * 4 + 2 * HD * 2);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nsplit = A.nsplit, nkv = A.NKV, nq = A.NQ, ldq = A.QKV, ctx_stride = A.ctx_stride;
    const bool mine = (int)blockIdx.x < nsplit * nkv && !gb.dead;
    const bool real = mine && wid < NW;
    const int sp = blockIdx.x % nsplit, kvh = mine ? blockIdx.x / nsplit : 0, b = 0;
    const int p_new = A.pos[b];
    const int ctx = p_new + 1;
    int chunk = (ctx + nsplit - 1) / nsplit;
    chunk = (chunk + 31) & ~31;
    const int beg = sp * chunk, end = min(ctx, beg + chunk);
    const int len = max(end - beg, 0);
    const int nit = (len + 31) >> 5;
    bf16_t* kb = Lw.kc + (size_t)A.slots[b] * A.slot_stride + (size_t)kvh * A.kv_head_stride;
    bf16_t* vb = Lw.vc + (size_t)A.slots[b] * A.slot_stride + (size_t)kvh * A.kv_head_stride;
    auto slice8 = [&](int n) -> u32x4_t {
        const float* pp = qpart + (size_t)b * ldq + n;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < qks; k0 += 4) {
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
        float a = qpart[(size_t)b * ldq + n];
        for (int k2 = 1; k2 < qks; ++k2) a += qpart[((size_t)k2 * SK_ROWS + b) * ldq + n];
        return f2bf(a);
    };
    const bool owner = len > 0 && end == ctx;
    const int prow = (i >> 2) * 8 + (i & 3);
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    u32x4_t kr[2][4], vr[8];
    auto load_k = [&](int it) {
        const int P0 = beg + it * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = P0 + prow + 4 * t;
            const bool ok = p < end && p != p_new;
            const bf16_t* src = kb + (size_t)p * HD + g * 16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                kr[t][s4] = ok ? *reinterpret_cast<const u32x4_t*>(src + (s4 >> 1) * 64 + (s4 & 1) * 8) : zero4;
        }
    };
    auto load_v = [&](int it) {
        const int P0 = beg + it * 32;
        const bool vok = P0 + g * 8 < end;
        const bf16_t* vsrc = vb + (size_t)i * ctx_stride + P0 + g * 8;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            vr[dt] = vok ? *reinterpret_cast<const u32x4_t*>(vsrc + (size_t)dt * 16 * ctx_stride) : zero4;
    };
    // the cache rows of earlier tokens depend on nothing this step produced: they stream while the barrier in front of this phase is counted
    const bool early = (A.prefetch & 3) && !((gb.opt & 1) && wid == 0);
    if (real && wid < nit && early) { load_k(wid); load_v(wid); }
    bar_wait(gb, s_flag);
    if (gb.dead || !mine) return;
    if (real && wid < nit && !early) { load_k(wid); load_v(wid); }

    auto rope_head = [&](int col, u32x4_t (&out)[4]) {
        u32x4_t x[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x[s4] = slice8(col + (s4 >> 1) * 64 + g * 16 + (s4 & 1) * 8);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float4* cp = reinterpret_cast<const float4*>(A.cos_t + (size_t)p_new * (HD / 2) + g * 16 + hf * 8);
            const float4* sq = reinterpret_cast<const float4*>(A.sin_t + (size_t)p_new * (HD / 2) + g * 16 + hf * 8);
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
    u32x4_t qf[4];
    if (real && i < GQ) rope_head((kvh * GQ + i) * HD, qf);
    else {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = zero4;
    }
    if (owner) {       // workgroup-uniform
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
        if (tid >= NW * 64 - 128 && tid < NW * 64) {
            const int d = tid - (NW * 64 - 128);
            const bf16_t x = elem((nq + nkv + kvh) * HD + d);
            vb[(size_t)d * ctx_stride + p_new] = x;
            s_new[HD + d] = x;
        }
        __syncthreads();
    }
    float m = -1e30f, l = 0.f;
    f32x4_t acc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (real) {
        for (int it = wid; it < nit; it += NW) {
            const int P0 = beg + it * 32;
            if (p_new >= P0 && p_new < P0 + 32) {
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
                    S[t] = mfma16(__builtin_bit_cast(bf16x8_t, kr[t][s4]), __builtin_bit_cast(bf16x8_t, qf[s4]), S[t]);
            }
            if (it + NW < nit) load_k(it + NW);
            float sv[8];
            float mx = -1e30f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = P0 + g * 8 + e < end;
                sv[e] = ok ? S[e >> 2][e & 3] * A.scale : -1e30f;
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
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                acc[dt] *= a;
                acc[dt] = mfma16(__builtin_bit_cast(bf16x8_t, vr[dt]), __builtin_bit_cast(bf16x8_t, pf), acc[dt]);
            }
            if (it + NW < nit) load_v(it + NW);
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (i < GQ) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(&s_acc[wid][i][dt * 16 + g * 4]) = acc[dt];
            if (g == 0) { s_m[wid][i] = m; s_l[wid][i] = l; }
        }
    }
    __syncthreads();
    if (real) {
        const size_t base = (((size_t)b * nq + kvh * GQ) * nsplit + sp) * (HD + 2);
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
            if (nsplit == 1) { __hip_atomic_store(&A.dO[(size_t)(kvh * GQ + hq) * HD + d], f2bf(o / L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
            __hip_atomic_store(&A.attn_ws[base + (size_t)hq * nsplit * (HD + 2) + d], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(&A.attn_ws[base + (size_t)hq * nsplit * (HD + 2) + HD], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.attn_ws[base + (size_t)hq * nsplit * (HD + 2) + HD + 1], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (nsplit == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(&A.tickets[b * nkv + kvh], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsplit - 1));
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    if (real) {
        for (int hq = wid; hq < GQ; hq += NW) {
            const float* w = A.attn_ws + (((size_t)b * nq + kvh * GQ + hq) * nsplit) * (HD + 2);
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
            __hip_atomic_store(reinterpret_cast<uint32_t*>(A.dO + (size_t)(kvh * GQ + hq) * HD + 2 * lane), pack2bf(num0 * inv, num1 * inv), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) __hip_atomic_store(&A.tickets[b * nkv + kvh], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(512) void decode_b1_persistent_kernel(B1Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_flag;
    __shared__ float s_ss[8];
    GridBar gb{A.bar, A.err, 0u, (int)gridDim.x, false, A.prefetch >> 2};
    bar_begin(gb, &s_flag);
    const unsigned final_epoch = gb.epoch + 5u * (unsigned)A.NL - 1u;     // five barriers per layer, none behind the last phase (its consumer is the next kernel):
                                                                          // every bar_arrive has its bar_wait, so the counters of all three levels agree at the end
    bf16_t *xa = A.xa, *xb = A.xb;
    for (int l = 0; l < A.NL; ++l) {
        const B1Layer Lw = A.layers[l];
        // qkv GEMV: sums the previous layer's down partials + residual, input norm
        {
            const Pro pro{l ? A.ws : nullptr, l ? A.pd.KS : 0, xa, xb, Lw.rms1, A.eps};
            gemv_phase<1, 1>(smem, A.pq, nullptr, Lw.wqkv, A.QKV, A.H, A.ws2, pro, gb, l > 0, A.prefetch, &s_flag, s_ss);
            bf16_t* t = xa; xa = xb; xb = t;
            bar_arrive(gb);
        }
        attn_phase(smem, A, Lw, A.ws2, A.pq.KS, gb, &s_flag);
        bar_arrive(gb);
        {
            const Pro pro{};
            gemv_phase<1, 0>(smem, A.po, A.dO, Lw.wo, A.H, A.H, A.ws, pro, gb, true, A.prefetch, &s_flag, s_ss);
            bar_arrive(gb);
        }
        {
            const Pro pro{A.ws, A.po.KS, xa, xb, Lw.rms2, A.eps};
            if (A.pg.NT == 2) gemv_phase<2, 1>(smem, A.pg, nullptr, Lw.wgu, 2 * A.I, A.H, A.ws2, pro, gb, true, A.prefetch, &s_flag, s_ss);
            else gemv_phase<1, 1>(smem, A.pg, nullptr, Lw.wgu, 2 * A.I, A.H, A.ws2, pro, gb, true, A.prefetch, &s_flag, s_ss);
            bf16_t* t = xa; xa = xb; xb = t;
            bar_arrive(gb);
        }
        {
            const Pro pro{A.ws2, A.pg.KS, nullptr, nullptr, nullptr, 0.f};
            gemv_phase<1, 2>(smem, A.pd, nullptr, Lw.wd, A.H, A.I, A.ws, pro, gb, true, A.prefetch, &s_flag, s_ss);
            if (l + 1 < A.NL) bar_arrive(gb);
        }
        if (gb.dead) return;
    }
    bar_end(gb, final_epoch);          // (workgroup 0 is past the last bar_wait: every workgroup has read BASE)
}

}  // namespace

size_t decode_b1_bar_bytes() { return gridbar::bar_bytes(); }
int decode_b1_num_cus() {          // CUs of the current device (cached per device id)
    static std::atomic<int> cache[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return -1;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cache[dev].store(prop.multiProcessorCount, std::memory_order_relaxed);
    return prop.multiProcessorCount;
}

int launch_decode_b1_persistent(const DecodeB1Args& a, hipStream_t s) {
    const int ncu = decode_b1_num_cus();
    if (ncu <= 0) return TRACE_ERR_HIP;
    const DecodeB1Plan* pl[4] = {&a.pq, &a.po, &a.pg, &a.pd};
    size_t lds = 0;
    int maxgrid = a.nsplit * a.NKV;
    for (int i = 0; i < 4; ++i) {
        const int NT = pl[i]->NT;
        if (NT != 1 && !(i == 2 && NT == 2)) return TRACE_ERR_STATE;      // two tiles per task: the gate|up product only
        const size_t need = (size_t)pl[i]->chunk_units * 2 * 1024 + (pl[i]->WPT > 1 ? (size_t)pl[i]->T * pl[i]->WPT * NT * 1024 : 0);
        lds = need > lds ? need : lds;
        if (pl[i]->threads > 512) return TRACE_ERR_STATE;
        maxgrid = pl[i]->grid > maxgrid ? pl[i]->grid : maxgrid;
    }
    const size_t attn_lds = 4 * 4 * 128 * 4 + 2 * 4 * 4 * 4 + 2 * 128 * 2 + 16;
    lds = lds > attn_lds ? lds : attn_lds;
    // every phase must fit the one-workgroup-per-CU grid (all workgroups resident: the grid barrier needs them), and the fused-norm prologue its 2 groups per thread
    if (maxgrid > ncu || (a.H >> 3) > 2 * a.pq.threads || (a.H >> 3) > 2 * a.pg.threads || a.nsplit < 1 || !a.bar || !a.err || !a.layers) return TRACE_ERR_STATE;
    static LdsGrantSized grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(decode_b1_persistent_kernel), lds)) return TRACE_ERR_HIP;
    hipLaunchKernelGGL(decode_b1_persistent_kernel, dim3(ncu), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
