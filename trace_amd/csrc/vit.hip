// CLIP ViT front end: patch extraction for the stride-14 conv-as-GEMM, and the embedding assembly
// (CLS + position embedding) fused with `pre_layrnorm`.
// Reference: HF CLIPVisionEmbeddings (conv k=s=P, no bias; cat CLS; + position_embedding) then
// CLIPVisionTransformer.pre_layrnorm — reached from trace/model/multimodal_encoder/clip_encoder.py:50.
#include "common.h"
#include "kernels.h"

namespace {

// One thread per (patch row, 8-wide k chunk): gathers 8 pixels (k = c*P*P + py*P + px), writes 16 bytes.
template <typename TIN>
__global__ __launch_bounds__(256) void im2col_kernel(const TIN* __restrict__ frames, bf16_t* __restrict__ A, int T, int S,
                                                     int P, int Kpad) {
    const int G = S / P, K = 3 * P * P, cpr = Kpad >> 3;
    const long total = (long)T * G * G * cpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / cpr;
        const int ch = (int)(idx - row * cpr);
        const int t = (int)(row / (G * G)), pr = (int)(row - (long)t * G * G);
        const int gy = pr / G, gx = pr - gy * G;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            float x = 0.f;
            if (k < K) {
                const int c = k / (P * P), rem = k - c * P * P;
                const int py = rem / P, px = rem - py * P;
                const size_t off = (((size_t)t * 3 + c) * S + (gy * P + py)) * S + gx * P + px;
                if (sizeof(TIN) == 2) x = bf2f(((const bf16_t*)frames)[off]);
                else x = ((const float*)frames)[off];
            }
            v[e] = x;
        }
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<uint4*>(A + row * Kpad + ch * 8) = o;
    }
}

constexpr int MAXCH = 8;
// One wave per output row X[t, j, :]: e = (j==0 ? cls : PE[t*GG + j-1]) + pos[j] (rounded to bf16 like the
// reference's bf16 embeddings), then LayerNorm(e) -> X.
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* __restrict__ PE, const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, const bf16_t* __restrict__ lw,
                                                           const bf16_t* __restrict__ lb, bf16_t* __restrict__ X, int T,
                                                           int GG, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int NT = GG + 1;
    if (row >= (long)T * NT) return;
    const int t = (int)(row / NT), j = (int)(row - (long)t * NT);
    const bf16_t* src = j == 0 ? cls : PE + ((size_t)t * GG + (j - 1)) * D;
    const bf16_t* ps = pos + (size_t)j * D;
    const int nch = D >> 3;
    float v[MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint4 a = *reinterpret_cast<const uint4*>(src + c * 8);
            const uint4 b = *reinterpret_cast<const uint4*>(ps + c * 8);
            const uint32_t aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][2 * e] = bf2f(f2bf(bflo(aa[e]) + bflo(bb[e])));
                v[i][2 * e + 1] = bf2f(f2bf(bfhi(aa[e]) + bfhi(bb[e])));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)D + eps);
    bf16_t* yr = X + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint4 uw = *reinterpret_cast<const uint4*>(lw + c * 8);
            const uint4 ub = *reinterpret_cast<const uint4*>(lb + c * 8);
            const uint32_t ww[4] = {uw.x, uw.y, uw.z, uw.w}, bb[4] = {ub.x, ub.y, ub.z, ub.w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = (v[i][2 * e] - mean) * rstd * bflo(ww[e]) + bflo(bb[e]);
                const float hi = (v[i][2 * e + 1] - mean) * rstd * bfhi(ww[e]) + bfhi(bb[e]);
                o[e] = pack2bf(lo, hi);
            }
            *reinterpret_cast<uint4*>(yr + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}
}  // namespace

int launch_im2col(const void* frames, int frames_fp32, bf16_t* A, int T, int S, int P, int Kpad, hipStream_t s) {
    if (S % P || Kpad % 8 || Kpad < 3 * P * P || T <= 0) return TRACE_ERR_ARG;
    const int G = S / P;
    const long total = (long)T * G * G * (Kpad / 8);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (frames_fp32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)frames, A, T, S, P, Kpad);
    else hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)frames, A, T, S, P, Kpad);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_vit_assemble(const bf16_t* PE, const bf16_t* cls, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb,
                        bf16_t* X, int T, int GG, int D, float eps, hipStream_t s) {
    if (D % 8 || D > MAXCH * 512 || T <= 0) return TRACE_ERR_ARG;
    const long rows = (long)T * (GG + 1);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, PE, cls, pos, lw, lb, X, T, GG, D, eps);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
