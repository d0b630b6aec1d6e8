// Mistral-side glue kernels (all HBM-bound row movers):
//   gather_rows  — embedding splice of prepare_inputs_labels_for_multimodal (reference trace/model/trace_arch.py:
//                  410-426 prefill; the per-frame [8 slots | 6 time tokens] interleave of :240-258)
//   rope_kv      — rotate-half RoPE on q (in place) and k (HF modeling_mistral apply_rotary_pos_emb), and the
//                  KV-cache append, for prefill rows or one decode row per sequence.
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(GatherTabs tabs, const int32_t* __restrict__ kind,
                                                          const int32_t* __restrict__ row, bf16_t* __restrict__ out,
                                                          int L, int H) {
    const int cpr = H >> 3;
    const long total = (long)L * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cpr), c = (int)(i - (long)r * cpr);
        const bf16_t* src = tabs.t[kind[r]] + (size_t)row[r] * H;
        *reinterpret_cast<uint4*>(out + (size_t)r * H + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
    }
}

// 8 lanes per (row, head): lane c handles pairs d = 8c..8c+7 with partner d + hd/2  (hd = 128)
__global__ __launch_bounds__(256) void rope_kv_kernel(bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ kcache,
                                                      bf16_t* __restrict__ vcache, long slot_stride, long kv_head_stride,
                                                      const int32_t* __restrict__ slot_arr, const int32_t* __restrict__ pos_arr,
                                                      int slot0, int pos0, int R, int nq, int nkv, int hd,
                                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t, int seq_len) {
    const int half = hd >> 1, lph = half >> 3;           // lanes per head
    const int nh = nq + 2 * nkv;
    const long total = (long)R * nh * lph;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % lph);
        const long rh = i / lph;
        const int hh = (int)(rh % nh), r = (int)(rh / nh);
        // rows are either one decode row per sequence (slot_arr / pos_arr), or `R / seq_len` equal-length prefill sequences
        // laid end to end going to consecutive slots
        const int sq = seq_len > 0 ? r / seq_len : 0;
        const int pos = pos_arr ? pos_arr[r] : pos0 + r - sq * seq_len;
        const int slot = slot_arr ? slot_arr[r] : slot0 + sq;
        bf16_t* x = qkv + (size_t)r * ld + (size_t)hh * hd;
        const uint4 u1 = *reinterpret_cast<const uint4*>(x + c * 8);
        const uint4 u2 = *reinterpret_cast<const uint4*>(x + half + c * 8);
        if (hh >= nq + nkv) {   // V: plain copy into a row-major cache (the engine's V^T cache is written by transpose_v / attn_decode)
            if (!vcache) continue;
            bf16_t* dst = vcache + (size_t)slot * slot_stride + (size_t)(hh - nq - nkv) * kv_head_stride + (size_t)pos * hd;
            *reinterpret_cast<uint4*>(dst + c * 8) = u1;
            *reinterpret_cast<uint4*>(dst + half + c * 8) = u2;
            continue;
        }
        const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)pos * half + c * 8);
        const float4* sp = reinterpret_cast<const float4*>(sin_t + (size_t)pos * half + c * 8);
        const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t a[4] = {u1.x, u1.y, u1.z, u1.w}, b[4] = {u2.x, u2.y, u2.z, u2.w};
        uint32_t o1[4], o2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x1l = bflo(a[e]), x1h = bfhi(a[e]), x2l = bflo(b[e]), x2h = bfhi(b[e]);
            // x*cos + rotate_half(x)*sin: first half x1*cos - x2*sin, second half x2*cos + x1*sin
            o1[e] = pack2bf(x1l * cs[2 * e] - x2l * sn[2 * e], x1h * cs[2 * e + 1] - x2h * sn[2 * e + 1]);
            o2[e] = pack2bf(x2l * cs[2 * e] + x1l * sn[2 * e], x2h * cs[2 * e + 1] + x1h * sn[2 * e + 1]);
        }
        bf16_t* dst = x;
        if (hh >= nq) dst = kcache + (size_t)slot * slot_stride + (size_t)(hh - nq) * kv_head_stride + (size_t)pos * hd;
        *reinterpret_cast<uint4*>(dst + c * 8) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<uint4*>(dst + half + c * 8) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    }
}
}  // namespace

int launch_gather_rows(const GatherTabs& tabs, const int32_t* kind, const int32_t* row, bf16_t* out, int L, int H,
                       hipStream_t s) {
    if (L <= 0 || H % 8) return TRACE_ERR_ARG;
    const long total = (long)L * (H / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, s, tabs, kind, row, out, L, H);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_rope_kv(bf16_t* qkv, int ld, bf16_t* kcache, bf16_t* vcache, long slot_stride, long kv_head_stride,
                   const int32_t* slot_arr, const int32_t* pos_arr, int slot0, int pos0, int R, int nq, int nkv, int hd,
                   const float* cos_t, const float* sin_t, int seq_len, hipStream_t s) {
    if (R <= 0 || hd % 16 || (ld % 8)) return TRACE_ERR_ARG;
    const long total = (long)R * (nq + 2 * nkv) * (hd / 16);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rope_kv_kernel, dim3(grid), dim3(256), 0, s, qkv, ld, kcache, vcache, slot_stride, kv_head_stride,
                       slot_arr, pos_arr, slot0, pos0, R, nq, nkv, hd, cos_t, sin_t, seq_len);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
