// Decode-step kernels (1 new token for each of B <= 16 sequences).  This is the HBM-bound heart of the path:
// every step streams all 7.2 B bf16 weights once (reference: TraceMistralForCausalLM.forward with
// input_ids [B,1] + past_key_values, trace/model/language_model/trace_mistral.py:114-264).
//
// skinny_gemm:  out[b,n] = sum_k X[b,k] W[n,k].  One workgroup = 16 weight rows (32 for the fused
//   gate|up pair), 8 waves split K; each lane streams 32 contiguous bytes of its weight row per step with
//   non-temporal 16-byte loads (a 16-lane row group covers full 128-byte lines), feeds them to the 16x16x32
//   bf16 MFMA as the A operand against the (L2-resident) activations as B, so B = 1..16 cost the same weight
//   stream.  The k-slot permutation trick (A and B fragments only have to agree on which k each slot means)
//   is what lets each lane read contiguous memory.  Partial tiles are combined through LDS.
// attn_decode:  single-query GQA attention over the KV cache, split over the context, + combine.
// head_logits / select_next:  active-head GEMV, masked arg-max (trace_mistral.py:244-252 + HF greedy), the
//   head-switch state machine (trace_mistral.py:86-88,336-344) and the next-token embedding
//   (trace_arch.py:345-375) — all on device, so a decode step never returns to the host.
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
// NB = number of 16-row activation groups (1: B <= 16, 2: B <= 32): the weight fragment is reused for both.
template <int EPI, int NB>
__global__ __launch_bounds__(512, 4) void skinny_gemm_kernel(const bf16_t* __restrict__ X, int ldx,
                                                                            const bf16_t* __restrict__ W, int ldw,
                                                                            bf16_t* __restrict__ out, int ldo,
                                                                            const bf16_t* __restrict__ R, int ldr, int B, int N, int K) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;      // 16-row weight tiles per workgroup
    constexpr int UN = NB == 1 ? 2 : 1;                  // 64-wide k units per load batch (keeps <= 128 VGPRs: 2 workgroups/CU)
    __shared__ float red[8][NT * NB][256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int U = K >> 6;
    const int u0 = (wid * U) >> 3, u1 = ((wid + 1) * U) >> 3;

    const bf16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + (size_t)(n0 + t * 16 + r) * ldw + g * 16;
    bool xon[NB];
    const bf16_t* xp[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        xon[nb] = r + 16 * nb < B;
        xp[nb] = X + (size_t)(xon[nb] ? r + 16 * nb : 0) * ldx + g * 16;
    }

    f32x4_t acc[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    Frag wa[UN][NT][2], wb[UN][NT][2], xa[UN][NB][2], xb[UN][NB][2];
    auto load = [&](Frag (&wf)[UN][NT][2], Frag (&xf)[UN][NB][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const bool ok = u + j < u1;
            const int ko = (u + j) * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wf[j][t][0].u = ok ? ldg_nt(wp[t] + ko) : make_uint4(0, 0, 0, 0);
                wf[j][t][1].u = ok ? ldg_nt(wp[t] + ko + 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                xf[j][nb][0].u = (ok && xon[nb]) ? *reinterpret_cast<const uint4*>(xp[nb] + ko) : make_uint4(0, 0, 0, 0);
                xf[j][nb][1].u = (ok && xon[nb]) ? *reinterpret_cast<const uint4*>(xp[nb] + ko + 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto mma = [&](Frag (&wf)[UN][NT][2], Frag (&xf)[UN][NB][2]) {
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][t][0].v, xf[j][nb][0].v, acc[t][nb], 0, 0, 0);
                    acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][t][1].v, xf[j][nb][1].v, acc[t][nb], 0, 0, 0);
                }
    };
    if (u0 < u1) {
        load(wa, xa, u0);
        for (int u = u0; u < u1; u += 2 * UN) {
            if (u + UN < u1) load(wb, xb, u + UN);
            mma(wa, xa);
            if (u + UN < u1) {
                if (u + 2 * UN < u1) load(wa, xa, u + 2 * UN);
                mma(wb, xb);
            }
        }
    }
    // acc[t][nb][i] = partial out[m = 16*nb + r][n = n0 + t*16 + g*4 + i]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wid][t * NB + nb][i * 64 + lane] = acc[t][nb][i];
    __syncthreads();
    if (tid < 256) {
        const int i = tid >> 6, l = tid & 63;
        const int nl = (l >> 4) * 4 + i;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int m = (l & 15) + 16 * nb;
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s += red[w][t * NB + nb][tid];
                v[t] = s;
            }
            if (m < B) {
                if (EPI == EPI_SWIGLU) {
                    const float gt = v[0], up = v[NT - 1];
                    out[(size_t)m * ldo + (n0 >> 1) + nl] = f2bf(gt / (1.f + __expf(-gt)) * up);
                } else {
                    float o = v[0];
                    if (EPI == EPI_RESIDUAL) o = bf2f(f2bf(o)) + bf2f(R[(size_t)m * ldr + n0 + nl]);
                    out[(size_t)m * ldo + n0 + nl] = f2bf(o);
                }
            }
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
__global__ __launch_bounds__(256, 3) void attn_decode_kernel(const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ kcache,
                                                             bf16_t* __restrict__ vtcache, long slot_stride, long kv_head_stride,
                                                             int ctx_stride, const int32_t* __restrict__ slots,
                                                             const int32_t* __restrict__ pos, float* __restrict__ ws,
                                                             unsigned int* __restrict__ tickets, bf16_t* __restrict__ O, int ldo,
                                                             int nq, int nkv, int nsplit, float scale, int fuse_rope,
                                                             const float* __restrict__ cos_t, const float* __restrict__ sin_t, int dbg) {
    constexpr int HD = 128, GQ = 4;
    __shared__ __attribute__((aligned(16))) float s_acc[4][GQ][HD];
    __shared__ float s_m[4][GQ], s_l[4][GQ];
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
    const bf16_t* row = qkv + (size_t)b * ldq;
    const bool owner = fuse_rope && len > 0 && end == ctx;       // this split holds the newest position

    const int prow = (i >> 2) * 8 + (i & 3);          // + 4 t: position (within the block) of A-row i of S tile t
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    u32x4_t kr[2][4], vr[8];
    auto load_k = [&](int it) {
        const int P0 = beg + it * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = P0 + prow + 4 * t;
            const bool ok = p < end && !(fuse_rope && p == p_new);
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
    if (wid < nit) { load_k(wid); load_v(wid); }      // cache blocks start streaming before anything else

    // rotate-half RoPE of a head's four 8-wide slices held by this lane: d = sp*64 + g*16 + hf*8 + e (its partner
    // d +- 64 is the same lane's other sp), rounded to bf16 like the stored q / k
    auto rope_head = [&](const bf16_t* head, u32x4_t (&out)[4]) {
        u32x4_t x[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x[s4] = *reinterpret_cast<const u32x4_t*>(head + (s4 >> 1) * 64 + g * 16 + (s4 & 1) * 8);
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
    if (i < GQ) rope_head(row + (size_t)(kvh * GQ + i) * HD, qf);
    else {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = zero4;
    }
    if (owner) {       // the newest row: append to the caches (K row-major, V down a column of V^T) and park it in LDS
        if (wid == 0 && i == 0) {
            u32x4_t kn[4];
            rope_head(row + (size_t)(nq + kvh) * HD, kn);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int d0 = (s4 >> 1) * 64 + g * 16 + (s4 & 1) * 8;
                *reinterpret_cast<u32x4_t*>(kb + (size_t)p_new * HD + d0) = kn[s4];
                *reinterpret_cast<u32x4_t*>(&s_new[d0]) = kn[s4];
            }
        }
        if (tid >= 128) {
            const bf16_t x = row[(size_t)(nq + nkv + kvh) * HD + tid - 128];
            vb[(size_t)(tid - 128) * ctx_stride + p_new] = x;
            s_new[HD + tid - 128] = x;
        }
        __syncthreads();                              // workgroup-uniform branch
    }

    float m = -1e30f, l = 0.f;                        // running max (per head = per column i) and this lane's partial sum
    f32x4_t acc[8];                                   // O^T tile dt: lane (i, g) reg r  <->  d = dt*16 + g*4 + r, head i
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // single register set, refilled as soon as the MFMAs that read it have issued: the next block's K streams in
    // during this block's softmax + PV, its V^T during the next block's QK (3 waves/SIMD cover the rest)
    for (int it = wid; it < nit; it += 4) {
        const int P0 = beg + it * 32;
        if (fuse_rope && p_new >= P0 && p_new < P0 + 32) {          // wave-uniform: splice the newest k / v (parked in LDS)
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
                S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kr[t][s4]),
                                                               __builtin_bit_cast(bf16x8_t, qf[s4]), S[t], 0, 0, 0);
        }
        if (it + 4 < nit) load_k(it + 4);
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
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            acc[dt] *= a;
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vr[dt]),
                                                              __builtin_bit_cast(bf16x8_t, pf), acc[dt], 0, 0, 0);
        }
        if (it + 4 < nit) load_v(it + 4);
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
        for (int x = tid; x < GQ * HD; x += 256) {
            const int hq = x >> 7, d = x & 127;
            const float M = fmaxf(fmaxf(s_m[0][hq], s_m[1][hq]), fmaxf(s_m[2][hq], s_m[3][hq]));
            float o = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
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
    {   // wave w merges q-head w: lane owns d = 2*lane, 2*lane+1; split loads are independent -> issued in batches
        const int hq = wid;
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
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[jj][0].v, x0[jj][0].v, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[jj][1].v, x0[jj][1].v, acc0, 0, 0, 0);
            if (NB == 2) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[jj][0].v, x1[jj][0].v, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[jj][1].v, x1[jj][1].v, acc1, 0, 0, 0);
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

// one workgroup; sequences handled one after another by all 256 threads
__global__ __launch_bounds__(256) void select_next_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
                                                          StepState st, const bf16_t* __restrict__ embed,
                                                          const bf16_t* __restrict__ time_tab, const bf16_t* __restrict__ score_tab,
                                                          const bf16_t* __restrict__ sync_row, bf16_t* __restrict__ xnext, int ldx,
                                                          int B, int H, int V, int Tv, int Sv, int ntiles, int advance) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_feed;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int step = *st.step;
    const int max_new = st.params[0], eos = st.params[1], record_feed = st.params[2];
    for (int b = 0; b < B; ++b) {
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
            if (advance) st.pos[b] += 1;
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
        __syncthreads();
    }
    if (tid == 0) *st.step = step + 1;
}
}  // namespace

int launch_skinny_gemm(const bf16_t* X, int ldx, const bf16_t* W, int ldw, bf16_t* out, int ldo, const bf16_t* R, int ldr,
                       int B, int N, int K, int epi, hipStream_t s) {
    if (B < 1 || B > 32 || K % 64 || (ldx % 8) || (ldw % 8)) return TRACE_ERR_ARG;
#define SK(EPI_, NB_, GRID_) hipLaunchKernelGGL((skinny_gemm_kernel<EPI_, NB_>), dim3(GRID_), dim3(512), 0, s, X, ldx, W, ldw, out, ldo, R, ldr, B, N, K)
    switch (epi) {
        case EPI_NONE:
            if (N % 16) return TRACE_ERR_ARG;
            if (B <= 16) SK(EPI_NONE, 1, N / 16); else SK(EPI_NONE, 2, N / 16);
            break;
        case EPI_RESIDUAL:
            if (N % 16 || !R) return TRACE_ERR_ARG;
            if (B <= 16) SK(EPI_RESIDUAL, 1, N / 16); else SK(EPI_RESIDUAL, 2, N / 16);
            break;
        case EPI_SWIGLU:
            if (N % 32) return TRACE_ERR_ARG;
            if (B <= 16) SK(EPI_SWIGLU, 1, N / 32); else SK(EPI_SWIGLU, 2, N / 32);
            break;
        default: return TRACE_ERR_ARG;
    }
#undef SK
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int g_attn_debug = 0;   // microbenchmark-only phase cut-offs (0 = full kernel)
int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* kcache, bf16_t* vtcache, long slot_stride, long kv_head_stride,
                       int ctx_stride, const int32_t* slots, const int32_t* pos, bf16_t* O, int ldo, float* ws, unsigned int* tickets, int B,
                       int nq, int nkv, int hd, int nsplit, float scale, int fuse_rope, const float* cos_t, const float* sin_t,
                       hipStream_t s) {
    if (hd != 128 || nq != 4 * nkv || nsplit < 1 || B < 1 || !tickets || ctx_stride % 32) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nsplit, nkv, B), dim3(256), 0, s, qkv, ldq, kcache, vtcache, slot_stride,
                       kv_head_stride, ctx_stride, slots, pos, ws, tickets, O, ldo, nq, nkv, nsplit, scale, fuse_rope, cos_t, sin_t,
                       g_attn_debug);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_head_logits(const bf16_t* X, int ldx, const bf16_t* Wh, int H, const int32_t* heads, int V, int Tv, int Sv,
                       float* part_val, int32_t* part_idx, float* logits_out, int B, hipStream_t s) {
    if (B < 1 || B > 32 || H % 64) return TRACE_ERR_ARG;
    const int ntiles = (V + 1 + Tv + Sv + 15) / 16;
    hipLaunchKernelGGL(head_logits_kernel, dim3(ntiles), dim3(512), 0, s, X, ldx, Wh, H, heads, V, Tv, Sv, part_val, part_idx,
                       logits_out, B, ntiles);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_select_next(const float* part_val, const int32_t* part_idx, const StepState& st, const bf16_t* embed,
                       const bf16_t* time_tab, const bf16_t* score_tab, const bf16_t* sync_row, bf16_t* xnext, int ldx, int B,
                       int H, int V, int Tv, int Sv, int advance, hipStream_t s) {
    if (B < 1 || B > 32 || H % 8) return TRACE_ERR_ARG;
    const int ntiles = (V + 1 + Tv + Sv + 15) / 16;
    hipLaunchKernelGGL(select_next_kernel, dim3(1), dim3(256), 0, s, part_val, part_idx, st, embed, time_tab, score_tab,
                       sync_row, xnext, ldx, B, H, V, Tv, Sv, ntiles, advance);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
