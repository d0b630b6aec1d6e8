// SpatialSlotPool (reference trace/model/multimodal_projector/builder.py:411-467), everything before `readout`:
//   x = LayerNorm_eps1e-6(feats)           (timm LayerNorm, :451)
//   x = x*cos + rotate_half(x)*sin         (channel-axis RoPE, position = patch index, :289-359,:453-455)
//   logits[n,s] = x[n,:] . slots[:,s]      (:457)      P = softmax over the n patches (:458)
//   res[s,:] = sum_n P[n,s] * x[n,:]       (:462)
// HBM-bound (one read of the 576x1024 frame features); one workgroup per frame, one wave per patch row.
// Pass 1 computes LN statistics + logits (slot matrix in LDS in lane-major order: conflict-free b128 reads),
// the 576-way softmax runs on the LDS logits, pass 2 re-reads the (L2-resident) row, re-applies LN+RoPE from
// the saved statistics and accumulates the 8 x 1024 weighted sums in registers; the waves add them into LDS one
// after the other (fixed order: bit-reproducible).  All arithmetic fp32; output bf16 [T*S, D], consumed by the readout GEMM.
#include "common.h"
#include "kernels.h"

namespace {
constexpr int NS = 8;   // slots

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    v[0] = bflo(u.x); v[1] = bfhi(u.x); v[2] = bflo(u.y); v[3] = bfhi(u.y);
    v[4] = bflo(u.z); v[5] = bfhi(u.z); v[6] = bflo(u.w); v[7] = bfhi(u.w);
}

__global__ __launch_bounds__(256) void slot_pool_kernel(const bf16_t* __restrict__ feats, long frame_stride, int row_stride,
                                                        const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b,
                                                        const bf16_t* __restrict__ slots, const float* __restrict__ cos_t,
                                                        const float* __restrict__ sin_t, bf16_t* __restrict__ res, int n,
                                                        int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NL = D >> 4;                       // active lanes per wave: lane owns d in [8l,8l+8) and D/2 + [8l,8l+8)
    const int H2 = D >> 1;
    // LDS carve-up
    uint4* s_slots = reinterpret_cast<uint4*>(smem);                       // [16][NL] x (8 slots bf16) = D*16 B
    float* s_logit = reinterpret_cast<float*>(smem + (size_t)D * 16);      // [n][8]
    float* s_stat = s_logit + (size_t)n * NS;                              // [n][2] mean, rstd
    float* s_res = s_stat + (size_t)n * 2;                                 // [8][D]
    float* s_red = s_res + (size_t)NS * D;                                 // [8][2] max, 1/sum

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t = blockIdx.x;
    const bf16_t* fr = feats + (size_t)t * frame_stride;

    for (int i = tid; i < D; i += 256) {
        // element d -> (e, owner lane): first half e = d&7, second half e = 8 + (d&7)
        const int half = i >= H2, dd = half ? i - H2 : i;
        const int ow = dd >> 3, e = (dd & 7) + 8 * half;
        s_slots[e * NL + ow] = *reinterpret_cast<const uint4*>(slots + (size_t)i * NS);
    }
    for (int i = tid; i < NS * D; i += 256) s_res[i] = 0.f;
    __syncthreads();

    const bool on = lane < NL;
    float w1[8], w2[8], b1[8], b2[8];
    if (on) {
        uint4 u;
        u = *reinterpret_cast<const uint4*>(ln_w + lane * 8); unpack8(u, w1);
        u = *reinterpret_cast<const uint4*>(ln_w + H2 + lane * 8); unpack8(u, w2);
        u = *reinterpret_cast<const uint4*>(ln_b + lane * 8); unpack8(u, b1);
        u = *reinterpret_cast<const uint4*>(ln_b + H2 + lane * 8); unpack8(u, b2);
    }

    // ---------------- pass 1: LN stats + RoPE + logits ----------------
    for (int p = wid; p < n; p += 4) {
        float x1[8], x2[8];
        float s = 0.f;
        if (on) {
            const bf16_t* xr = fr + (size_t)p * row_stride;
            uint4 u = *reinterpret_cast<const uint4*>(xr + lane * 8); unpack8(u, x1);
            u = *reinterpret_cast<const uint4*>(xr + H2 + lane * 8); unpack8(u, x2);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x1[e] + x2[e];
        }
        s = wave_sum(s);
        const float mean = s / (float)D;
        float q = 0.f;
        if (on) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float a = x1[e] - mean, b = x2[e] - mean; q += a * a + b * b; }
        }
        q = wave_sum(q);
        const float rstd = rsqrtf(q / (float)D + eps);
        float lg[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) lg[k] = 0.f;
        if (on) {
            const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)p * H2 + lane * 8);
            const float4* sp = reinterpret_cast<const float4*>(sin_t + (size_t)p * H2 + lane * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (x1[e] - mean) * rstd * w1[e] + b1[e];
                const float b = (x2[e] - mean) * rstd * w2[e] + b2[e];
                const float r1 = a * cs[e] - b * sn[e];
                const float r2 = b * cs[e] + a * sn[e];
                float sv[8];
                unpack8(s_slots[e * NL + lane], sv);
#pragma unroll
                for (int k = 0; k < NS; ++k) lg[k] += r1 * sv[k];
                unpack8(s_slots[(e + 8) * NL + lane], sv);
#pragma unroll
                for (int k = 0; k < NS; ++k) lg[k] += r2 * sv[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) lg[k] = wave_sum(lg[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) s_logit[p * NS + k] = lg[k];
            s_stat[p * 2] = mean;
            s_stat[p * 2 + 1] = rstd;
        }
    }
    __syncthreads();

    // ---------------- softmax over patches, per slot (2 slots per wave) ----------------
    for (int k = wid * 2; k < wid * 2 + 2; ++k) {
        float mx = -1e30f;
        for (int p = lane; p < n; p += 64) mx = fmaxf(mx, s_logit[p * NS + k]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int p = lane; p < n; p += 64) sm += __expf(s_logit[p * NS + k] - mx);
        sm = wave_sum(sm);
        if (lane == 0) { s_red[k * 2] = mx; s_red[k * 2 + 1] = 1.f / sm; }
    }
    __syncthreads();
    for (int i = tid; i < n * NS; i += 256) {
        const int k = i & (NS - 1);
        s_logit[i] = __expf(s_logit[i] - s_red[k * 2]) * s_red[k * 2 + 1];
    }
    __syncthreads();

    // ---------------- pass 2: res[s, d] = sum_p P[p, s] * x[p, d] ----------------
    float a1[NS][8], a2[NS][8];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[k][e] = 0.f; a2[k][e] = 0.f; }
    if (on) {
        for (int p = wid; p < n; p += 4) {
            float x1[8], x2[8];
            const bf16_t* xr = fr + (size_t)p * row_stride;
            uint4 u = *reinterpret_cast<const uint4*>(xr + lane * 8); unpack8(u, x1);
            u = *reinterpret_cast<const uint4*>(xr + H2 + lane * 8); unpack8(u, x2);
            const float mean = s_stat[p * 2], rstd = s_stat[p * 2 + 1];
            const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)p * H2 + lane * 8);
            const float4* sp = reinterpret_cast<const float4*>(sin_t + (size_t)p * H2 + lane * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float4 pa = *reinterpret_cast<const float4*>(s_logit + p * NS);
            const float4 pb = *reinterpret_cast<const float4*>(s_logit + p * NS + 4);
            const float pr[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (x1[e] - mean) * rstd * w1[e] + b1[e];
                const float b = (x2[e] - mean) * rstd * w2[e] + b2[e];
                const float r1 = a * cs[e] - b * sn[e];
                const float r2 = b * cs[e] + a * sn[e];
#pragma unroll
                for (int k = 0; k < NS; ++k) { a1[k][e] += pr[k] * r1; a2[k][e] += pr[k] * r2; }
            }
        }
    }
    // the four waves add their partial sums one after the other (LDS float atomics would add them in arrival order: the
    // result then differs in the last bit from run to run, which 32 decoder layers amplify to 0.2 in the prefill hidden state)
    for (int w = 0; w < 4; ++w) {
        if (on && wid == w) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s_res[k * D + lane * 8 + e] += a1[k][e];
                    s_res[k * D + H2 + lane * 8 + e] += a2[k][e];
                }
        }
        __syncthreads();
    }
    bf16_t* out = res + (size_t)t * NS * D;
    for (int i = tid; i < NS * D / 2; i += 256) {
        const float2 v = *reinterpret_cast<const float2*>(s_res + 2 * i);
        reinterpret_cast<uint32_t*>(out)[i] = pack2bf(v.x, v.y);
    }
}
}  // namespace

int launch_slot_pool(const bf16_t* feats, long frame_stride, int row_stride, const bf16_t* ln_w, const bf16_t* ln_b,
                     const bf16_t* slots, const float* cos_t, const float* sin_t, bf16_t* res, int T, int n, int D, int S,
                     float eps, hipStream_t s) {
    if (S != NS || D % 16 || D > 1024 || T <= 0 || n <= 0 || (row_stride % 8)) return TRACE_ERR_ARG;
    const size_t lds = (size_t)D * 16 + (size_t)n * NS * 4 + (size_t)n * 8 + (size_t)NS * D * 4 + 64;
    if (lds > 160 * 1024) return TRACE_ERR_ARG;
    static size_t set_for = 0;
    if (lds > set_for) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(slot_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        set_for = lds;
    }
    hipLaunchKernelGGL(slot_pool_kernel, dim3(T), dim3(256), lds, s, feats, frame_stride, row_stride, ln_w, ln_b, slots,
                       cos_t, sin_t, res, n, D, eps);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
