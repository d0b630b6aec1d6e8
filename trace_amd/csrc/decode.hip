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
// grid (nsplit, nkv, B), 256 threads; 16 lanes per cache row (16 B each = one 256-byte row per 16-lane group),
// every lane pre-loads its K and V rows up front so the two HBM round trips overlap.  Partials go to `ws`
// ([b][q-head][split][hd + 2] fp32) with write-through (sc1) stores; every wave drains vmcnt, then one lane takes
// an agent-scope ticket; the last-arriving workgroup of a (b, kv-head) pair does ONE agent-scope acquire and merges
// the splits (placement-independent: no assumption on dispatch order or XCD), then re-zeroes the ticket.
__global__ __launch_bounds__(256, 2) void attn_decode_kernel(const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ kcache,
                                                             bf16_t* __restrict__ vcache, long slot_stride, long kv_head_stride,
                                                             const int32_t* __restrict__ slots, const int32_t* __restrict__ pos,
                                                             float* __restrict__ ws, unsigned int* __restrict__ tickets,
                                                             bf16_t* __restrict__ O, int ldo, int nq, int nkv, int nsplit,
                                                             float scale, int fuse_rope, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, int dbg) {
    constexpr int HD = 128, GQ = 4, PF = 4;          // PF = cache rows per 16-lane group per prefetch batch
    __shared__ float s_acc[4][GQ][HD];
    __shared__ float s_m[4][GQ], s_l[4][GQ];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane >> 4, c = lane & 15;
    const int sp = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int p_new = pos[b];
    const int ctx = p_new + 1;
    int chunk = (ctx + nsplit - 1) / nsplit;
    chunk = (chunk + 15) & ~15;
    const int beg = sp * chunk, end = min(ctx, beg + chunk);
    const int len = max(end - beg, 0);
    const int nit = (len + 15) >> 4;                  // 16 rows per workgroup iteration (4 waves x 4 groups)
    bf16_t* kb = kcache + (size_t)slots[b] * slot_stride + (size_t)kvh * kv_head_stride;
    bf16_t* vb = vcache + (size_t)slots[b] * slot_stride + (size_t)kvh * kv_head_stride;

    uint4 kA[PF], vA[PF], kB[PF], vB[PF];
    auto load = [&](uint4 (&kr)[PF], uint4 (&vr)[PF], int it0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = (it0 + u) * 16 + wid * 4 + j;
            const bool ok = i < len && !(fuse_rope && beg + i == p_new);
            kr[u] = ok ? *reinterpret_cast<const uint4*>(kb + (size_t)(beg + i) * HD + c * 8) : make_uint4(0, 0, 0, 0);
            vr[u] = ok ? *reinterpret_cast<const uint4*>(vb + (size_t)(beg + i) * HD + c * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    if (nit > 0) load(kA, vA, 0);                     // cache rows start streaming before anything else

    // ---- q (4 heads of this kv group) and, when fused, RoPE + the new k/v row ----
    float qv[GQ][8];
    uint4 knew = make_uint4(0, 0, 0, 0), vnew = make_uint4(0, 0, 0, 0);
    {
        const bf16_t* row = qkv + (size_t)b * ldq;
        float cs[8], sn[8];
        if (fuse_rope) {
            const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)p_new * (HD / 2) + (c & 7) * 8);
            const float4* sq = reinterpret_cast<const float4*>(sin_t + (size_t)p_new * (HD / 2) + (c & 7) * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sq[0], s1 = sq[1];
            cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
            sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
        }
        const float sgn = c < 8 ? -1.f : 1.f;      // first half: x*cos - partner*sin ; second half: x*cos + partner*sin
        auto rot = [&](const bf16_t* head, float* out) {
            const uint4 u = *reinterpret_cast<const uint4*>(head + c * 8);
            const float x[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
            if (!fuse_rope) {
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = x[e];
                return;
            }
            const uint4 w = *reinterpret_cast<const uint4*>(head + (c ^ 8) * 8);
            const float y[8] = {bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y), bflo(w.z), bfhi(w.z), bflo(w.w), bfhi(w.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = bf2f(f2bf(x[e] * cs[e] + sgn * y[e] * sn[e]));   // bf16 like the stored q/k
        };
#pragma unroll
        for (int hq = 0; hq < GQ; ++hq) rot(row + (size_t)(kvh * GQ + hq) * HD, qv[hq]);
        if (fuse_rope) {
            float kn[8];
            rot(row + (size_t)(nq + kvh) * HD, kn);
            knew = make_uint4(pack2bf(kn[0], kn[1]), pack2bf(kn[2], kn[3]), pack2bf(kn[4], kn[5]), pack2bf(kn[6], kn[7]));
            vnew = *reinterpret_cast<const uint4*>(row + (size_t)(nq + nkv + kvh) * HD + c * 8);
            if (len > 0 && end == ctx && wid == 0 && j == 0) {      // the split that owns the newest row appends it to the cache
                *reinterpret_cast<uint4*>(kb + (size_t)p_new * HD + c * 8) = knew;
                *reinterpret_cast<uint4*>(vb + (size_t)p_new * HD + c * 8) = vnew;
            }
        }
#pragma unroll
        for (int hq = 0; hq < GQ; ++hq)
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[hq][e] *= scale;
    }
    // ---- online softmax per 16-lane group: running max m, sum l, and this lane's 8-wide slice of o, per q-head ----
    float m[GQ], l[GQ], acc[GQ][8];
#pragma unroll
    for (int hq = 0; hq < GQ; ++hq) {
        m[hq] = -1e30f; l[hq] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[hq][e] = 0.f;
    }
    auto process = [&](uint4 (&kr)[PF], uint4 (&vr)[PF], int it0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (it0 + u >= nit) break;
            const int i = (it0 + u) * 16 + wid * 4 + j;
            const bool fresh = fuse_rope && beg + i == p_new;
            const uint4 ku = fresh ? knew : kr[u];
            const uint4 vu = fresh ? vnew : vr[u];
            const float kv[8] = {bflo(ku.x), bfhi(ku.x), bflo(ku.y), bfhi(ku.y), bflo(ku.z), bfhi(ku.z), bflo(ku.w), bfhi(ku.w)};
            const float vv[8] = {bflo(vu.x), bfhi(vu.x), bflo(vu.y), bfhi(vu.y), bflo(vu.z), bfhi(vu.z), bflo(vu.w), bfhi(vu.w)};
#pragma unroll
            for (int hq = 0; hq < GQ; ++hq) {
                float sc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sc += qv[hq][e] * kv[e];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sc += __shfl_xor(sc, o, 64);
                if (i < len) {
                    const float mn = fmaxf(m[hq], sc);
                    const float a = __expf(m[hq] - mn), p = __expf(sc - mn);
                    m[hq] = mn;
                    l[hq] = l[hq] * a + p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[hq][e] = acc[hq][e] * a + p * vv[e];
                }
            }
        }
    };
    for (int it0 = 0; it0 < nit; it0 += 2 * PF) {
        if (it0 + PF < nit) load(kB, vB, it0 + PF);
        process(kA, vA, it0);
        if (it0 + PF < nit) {
            if (it0 + 2 * PF < nit) load(kA, vA, it0 + 2 * PF);
            process(kB, vB, it0 + PF);
        }
    }
    // ---- merge the 4 groups of the wave (xor 16, 32), then the 4 waves through LDS ----
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
        for (int hq = 0; hq < GQ; ++hq) {
            const float mo = __shfl_xor(m[hq], o, 64), lo = __shfl_xor(l[hq], o, 64);
            const float M = fmaxf(m[hq], mo);
            const float fa = __expf(m[hq] - M), fb = __expf(mo - M);
            l[hq] = l[hq] * fa + lo * fb;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hq][e] = acc[hq][e] * fa + __shfl_xor(acc[hq][e], o, 64) * fb;
            m[hq] = M;
        }
    }
    if (j == 0) {
#pragma unroll
        for (int hq = 0; hq < GQ; ++hq) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s_acc[wid][hq][c * 8 + e] = acc[hq][e];
            if (c == 0) { s_m[wid][hq] = m[hq]; s_l[wid][hq] = l[hq]; }
        }
    }
    __syncthreads();
    if (dbg == 2) return;
    {
        const size_t base = (((size_t)b * nq + kvh * GQ) * nsplit + sp) * (HD + 2);
        // write-through (sc1) stores: visible at agent scope once vmcnt drains, no L2 write-back fence needed
        for (int i = tid; i < GQ * HD; i += 256) {
            const int hq = i >> 7, d = i & 127;
            const float M = fmaxf(fmaxf(s_m[0][hq], s_m[1][hq]), fmaxf(s_m[2][hq], s_m[3][hq]));
            float o = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = __expf(s_m[w][hq] - M);
                o += f * s_acc[w][hq][d];
                L += f * s_l[w][hq];
            }
            __hip_atomic_store(&ws[base + (size_t)hq * nsplit * (HD + 2) + d], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(&ws[base + (size_t)hq * nsplit * (HD + 2) + HD], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ws[base + (size_t)hq * nsplit * (HD + 2) + HD + 1], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (dbg == 1) return;
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
int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* kcache, bf16_t* vcache, long slot_stride, long kv_head_stride,
                       const int32_t* slots, const int32_t* pos, bf16_t* O, int ldo, float* ws, unsigned int* tickets, int B,
                       int nq, int nkv, int hd, int nsplit, float scale, int fuse_rope, const float* cos_t, const float* sin_t,
                       hipStream_t s) {
    if (hd != 128 || nq != 4 * nkv || nsplit < 1 || B < 1 || !tickets) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nsplit, nkv, B), dim3(256), 0, s, qkv, ldq, kcache, vcache, slot_stride,
                       kv_head_stride, slots, pos, ws, tickets, O, ldo, nq, nkv, nsplit, scale, fuse_rope, cos_t, sin_t,
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
