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
template <int EPI>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ W,
                                                          int ldw, bf16_t* __restrict__ out, int ldo,
                                                          const bf16_t* __restrict__ R, int ldr, int B, int N, int K) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;      // 16-row weight tiles per workgroup
    constexpr int UN = 2;                                // 64-wide k units per batch (4 x 16 B per lane per tile)
    __shared__ float red[8][NT][256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int U = K >> 6;
    const int u0 = (wid * U) >> 3, u1 = ((wid + 1) * U) >> 3;

    const bf16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + (size_t)(n0 + t * 16 + r) * ldw + g * 16;
    const bool xon = r < B;
    const bf16_t* xp = X + (size_t)(xon ? r : 0) * ldx + g * 16;

    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    Frag wa[UN][NT][2], wb[UN][NT][2], xa[UN][2], xb[UN][2];
    auto load = [&](Frag (&wf)[UN][NT][2], Frag (&xf)[UN][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const bool ok = u + j < u1;
            const int ko = (u + j) * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wf[j][t][0].u = ok ? ldg_nt(wp[t] + ko) : make_uint4(0, 0, 0, 0);
                wf[j][t][1].u = ok ? ldg_nt(wp[t] + ko + 8) : make_uint4(0, 0, 0, 0);
            }
            xf[j][0].u = (ok && xon) ? *reinterpret_cast<const uint4*>(xp + ko) : make_uint4(0, 0, 0, 0);
            xf[j][1].u = (ok && xon) ? *reinterpret_cast<const uint4*>(xp + ko + 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto mma = [&](Frag (&wf)[UN][NT][2], Frag (&xf)[UN][2]) {
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][t][0].v, xf[j][0].v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][t][1].v, xf[j][1].v, acc[t], 0, 0, 0);
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
    // acc[t][i] = partial out[m = r][n = n0 + t*16 + g*4 + i]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wid][t][i * 64 + lane] = acc[t][i];
    __syncthreads();
    if (tid < 256) {
        const int i = tid >> 6, l = tid & 63;
        const int m = l & 15, nl = (l >> 4) * 4 + i;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w][t][tid];
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

// ---------------------------------------------------------------------------------------------------------
// grid (nsplit, nkv, B); 256 threads.  ws layout per (b, q-head, split): [hd] o (unnormalised) then m, l.
constexpr int AD_MAXCHUNK = 1024;
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kcache,
                                                          const bf16_t* __restrict__ vcache, long slot_stride,
                                                          long kv_head_stride, const int32_t* __restrict__ slots,
                                                          const int32_t* __restrict__ pos, float* __restrict__ ws, int nq,
                                                          int nkv, int nsplit, float scale) {
    constexpr int HD = 128, GQ = 4;
    __shared__ float s_p[GQ][AD_MAXCHUNK];
    __shared__ float s_o[GQ][HD];
    __shared__ float s_ml[GQ][2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane >> 4, c = lane & 15;
    const int sp = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int ctx = pos[b] + 1;
    int chunk = (ctx + nsplit - 1) / nsplit;
    chunk = min((chunk + 15) & ~15, AD_MAXCHUNK);
    const int beg = sp * chunk, end = min(ctx, beg + chunk);
    const int len = max(end - beg, 0);
    const bf16_t* kb = kcache + (size_t)slots[b] * slot_stride + (size_t)kvh * kv_head_stride;
    const bf16_t* vb = vcache + (size_t)slots[b] * slot_stride + (size_t)kvh * kv_head_stride;
    for (int i = tid; i < GQ * HD; i += 256) (&s_o[0][0])[i] = 0.f;

    float qv[GQ][8];
#pragma unroll
    for (int hq = 0; hq < GQ; ++hq) {
        const uint4 u = *reinterpret_cast<const uint4*>(q + (size_t)b * ldq + (size_t)(kvh * GQ + hq) * HD + c * 8);
        qv[hq][0] = bflo(u.x); qv[hq][1] = bfhi(u.x); qv[hq][2] = bflo(u.y); qv[hq][3] = bfhi(u.y);
        qv[hq][4] = bflo(u.z); qv[hq][5] = bfhi(u.z); qv[hq][6] = bflo(u.w); qv[hq][7] = bfhi(u.w);
    }
    // ---- scores ----
    for (int i0 = wid * 4; i0 < len; i0 += 16) {
        const int i = i0 + j;
        float part[GQ] = {0.f, 0.f, 0.f, 0.f};
        if (i < len) {
            const uint4 u = *reinterpret_cast<const uint4*>(kb + (size_t)(beg + i) * HD + c * 8);
            const float kv[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
            for (int hq = 0; hq < GQ; ++hq)
#pragma unroll
                for (int e = 0; e < 8; ++e) part[hq] += qv[hq][e] * kv[e];
        }
#pragma unroll
        for (int hq = 0; hq < GQ; ++hq) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) part[hq] += __shfl_xor(part[hq], o, 64);
        }
        if (c == 0 && i < len) {
#pragma unroll
            for (int hq = 0; hq < GQ; ++hq) s_p[hq][i] = part[hq] * scale;
        }
    }
    __syncthreads();
    // ---- softmax over the chunk: wave w owns q-head w ----
    {
        float mx = -1e30f;
        for (int i = lane; i < len; i += 64) mx = fmaxf(mx, s_p[wid][i]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int i = lane; i < len; i += 64) {
            const float p = __expf(s_p[wid][i] - mx);
            s_p[wid][i] = p;
            sm += p;
        }
        sm = wave_sum(sm);
        if (lane == 0) { s_ml[wid][0] = mx; s_ml[wid][1] = sm; }
    }
    __syncthreads();
    // ---- o[hq, d] = sum_i p[hq, i] * V[i, d] ----
    float acc[GQ][8];
#pragma unroll
    for (int hq = 0; hq < GQ; ++hq)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[hq][e] = 0.f;
    for (int i0 = wid * 4; i0 < len; i0 += 16) {
        const int i = i0 + j;
        if (i < len) {
            const uint4 u = *reinterpret_cast<const uint4*>(vb + (size_t)(beg + i) * HD + c * 8);
            const float vv[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
            for (int hq = 0; hq < GQ; ++hq) {
                const float p = s_p[hq][i];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[hq][e] += p * vv[e];
            }
        }
    }
#pragma unroll
    for (int hq = 0; hq < GQ; ++hq)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[hq][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (j == 0) atomicAdd(&s_o[hq][c * 8 + e], v);
        }
    __syncthreads();
    for (int i = tid; i < GQ * HD; i += 256) {
        const int hq = i >> 7, d = i & 127;
        float* w = ws + (((size_t)b * nq + kvh * GQ + hq) * nsplit + sp) * (HD + 2);
        w[d] = s_o[hq][d];
        if (d == 0) { w[HD] = s_ml[hq][0]; w[HD + 1] = s_ml[hq][1]; }
    }
}

// grid (nq, B), 128 threads
__global__ __launch_bounds__(128) void attn_combine_kernel(const float* __restrict__ ws, bf16_t* __restrict__ O, int ldo, int nq,
                                                           int nsplit) {
    constexpr int HD = 128;
    const int hq = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* w = ws + ((size_t)b * nq + hq) * nsplit * (HD + 2);
    float M = -1e30f;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, w[s * (HD + 2) + HD]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float f = __expf(w[s * (HD + 2) + HD] - M);
        num += f * w[s * (HD + 2) + d];
        den += f * w[s * (HD + 2) + HD + 1];
    }
    O[(size_t)b * ldo + hq * HD + d] = f2bf(num / den);
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
    __shared__ float red[8][256];
    __shared__ float fin[256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int NV = V + 1 + Tv + Sv;
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
    const bool xon = r < B;
    const bf16_t* xp = X + (size_t)(xon ? r : 0) * ldx + g * 16;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int u = u0; u < u1; u += 4) {
        Frag w[4][2], x[4][2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const bool ok = u + jj < u1;
            const int ko = (u + jj) * 64;
            w[jj][0].u = ok ? ldg_nt(wp + ko) : make_uint4(0, 0, 0, 0);
            w[jj][1].u = ok ? ldg_nt(wp + ko + 8) : make_uint4(0, 0, 0, 0);
            x[jj][0].u = (ok && xon) ? *reinterpret_cast<const uint4*>(xp + ko) : make_uint4(0, 0, 0, 0);
            x[jj][1].u = (ok && xon) ? *reinterpret_cast<const uint4*>(xp + ko + 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[jj][0].v, x[jj][0].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[jj][1].v, x[jj][1].v, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wid][i * 64 + lane] = acc[i];
    __syncthreads();
    if (tid < 256) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[w][tid];
        // tid = i*64 + l  ->  m = l&15, n_local = (l>>4)*4 + i ; store as fin[m*16 + n_local]
        const int i = tid >> 6, l = tid & 63;
        fin[(l & 15) * 16 + (l >> 4) * 4 + i] = s;
    }
    __syncthreads();
    if (tid < B * 16) {
        const int b = tid >> 4, nl = tid & 15, n = n0 + nl;
        int lo, hi;
        head_bounds(heads[b], V, Tv, Sv, lo, hi);
        const bool ok = n >= lo && n < hi;
        float v = ok ? fin[b * 16 + nl] : -INFINITY;
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
    const int max_new = st.params[0], eos = st.params[1];
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
            const bool was_done = st.done[b] != 0;
            if (!was_done && step < max_new) {
                st.out_ids[(size_t)b * max_new + step] = tok;
                st.out_len[b] = step + 1;
                if (eos >= 0 && tok == eos) st.done[b] = 1;
            }
            int feed = tok;
            if (step < max_new) {
                const int f = st.forced[(size_t)b * max_new + step];
                if (f >= 0) feed = f;
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
    if (B < 1 || B > 16 || K % 64 || (ldx % 8) || (ldw % 8)) return TRACE_ERR_ARG;
    switch (epi) {
        case EPI_NONE:
            if (N % 16) return TRACE_ERR_ARG;
            hipLaunchKernelGGL(skinny_gemm_kernel<EPI_NONE>, dim3(N / 16), dim3(512), 0, s, X, ldx, W, ldw, out, ldo, R, ldr, B, N, K);
            break;
        case EPI_RESIDUAL:
            if (N % 16 || !R) return TRACE_ERR_ARG;
            hipLaunchKernelGGL(skinny_gemm_kernel<EPI_RESIDUAL>, dim3(N / 16), dim3(512), 0, s, X, ldx, W, ldw, out, ldo, R, ldr, B, N, K);
            break;
        case EPI_SWIGLU:
            if (N % 32) return TRACE_ERR_ARG;
            hipLaunchKernelGGL(skinny_gemm_kernel<EPI_SWIGLU>, dim3(N / 32), dim3(512), 0, s, X, ldx, W, ldw, out, ldo, R, ldr, B, N, K);
            break;
        default: return TRACE_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_attn_decode(const bf16_t* q, int ldq, const bf16_t* kcache, const bf16_t* vcache, long slot_stride,
                       long kv_head_stride, const int32_t* slots, const int32_t* pos, bf16_t* O, int ldo, float* ws, int B,
                       int nq, int nkv, int hd, int nsplit, float scale, hipStream_t s) {
    if (hd != 128 || nq != 4 * nkv || nsplit < 1 || B < 1) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nsplit, nkv, B), dim3(256), 0, s, q, ldq, kcache, vcache, slot_stride,
                       kv_head_stride, slots, pos, ws, nq, nkv, nsplit, scale);
    hipLaunchKernelGGL(attn_combine_kernel, dim3(nq, B), dim3(128), 0, s, ws, O, ldo, nq, nsplit);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_head_logits(const bf16_t* X, int ldx, const bf16_t* Wh, int H, const int32_t* heads, int V, int Tv, int Sv,
                       float* part_val, int32_t* part_idx, float* logits_out, int B, hipStream_t s) {
    if (B < 1 || B > 16 || H % 64) return TRACE_ERR_ARG;
    const int ntiles = (V + 1 + Tv + Sv + 15) / 16;
    hipLaunchKernelGGL(head_logits_kernel, dim3(ntiles), dim3(512), 0, s, X, ldx, Wh, H, heads, V, Tv, Sv, part_val, part_idx,
                       logits_out, B, ntiles);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_select_next(const float* part_val, const int32_t* part_idx, const StepState& st, const bf16_t* embed,
                       const bf16_t* time_tab, const bf16_t* score_tab, const bf16_t* sync_row, bf16_t* xnext, int ldx, int B,
                       int H, int V, int Tv, int Sv, int advance, hipStream_t s) {
    if (B < 1 || B > 16 || H % 8) return TRACE_ERR_ARG;
    const int ntiles = (V + 1 + Tv + Sv + 15) / 16;
    hipLaunchKernelGGL(select_next_kernel, dim3(1), dim3(256), 0, s, part_val, part_idx, st, embed, time_tab, score_tab,
                       sync_row, xnext, ldx, B, H, V, Tv, Sv, ntiles, advance);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
