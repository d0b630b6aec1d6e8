// STCConnector support kernels (reference trace/model/multimodal_projector/builder.py:138-249; legacy trace.infer()
// path).  Everything GEMM-shaped in the connector (1x1 convs, the 2x2x2 stride-2 Conv3d as an im2col GEMM, the SE
// fully-connected layers, the readout MLP) runs on the MFMA kernels of gemm.hip / decode.hip; what is left are
// HBM-bound channels-last row kernels: depthwise 3x3, global average pool, bias/activation, SE scaling,
// residual + SiLU, the Conv3d patch gather, and a one-off weight permute.  Layout: [frame][h][w][C] bf16.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    v[0] = bflo(u.x); v[1] = bfhi(u.x); v[2] = bflo(u.y); v[3] = bfhi(u.y);
    v[4] = bflo(u.z); v[5] = bfhi(u.z); v[6] = bflo(u.w); v[7] = bfhi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
__device__ __forceinline__ float act_fn(float x, int act) {
    if (act == ACT_SILU) return x / (1.f + __expf(-x));
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
    if (act == ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));     // nn.GELU() (exact erf form)
    return x;
}

// depthwise 3x3, pad 1, stride 1: w [C][9] (Conv2d weight [C,1,3,3]); one thread per (pixel, 8-channel chunk)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        bf16_t* __restrict__ y, int N, int H, int W, int C) {
    const int cpr = C >> 3;
    const long total = (long)N * H * W * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const long pix = i / cpr;
        const int px = (int)(pix % W), py = (int)((pix / W) % H);
        const long n = pix / ((long)W * H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // the 8 channels' 72 weights are 144 contiguous, 16-byte aligned bytes of w: nine 16-byte loads (round 5; the first version fetched them as 72
        // two-byte gathers per thread and ran at 199 us per [16, 24, 24, 4096] call — 0.75 TB/s — where the tensor moves in ~25; same sums in the same order)
        uint32_t wr[36];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const uint4 u = *reinterpret_cast<const uint4*>(w + (size_t)ch * 72 + q * 8);
            wr[4 * q] = u.x; wr[4 * q + 1] = u.y; wr[4 * q + 2] = u.z; wr[4 * q + 3] = u.w;
        }
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = py + dy, xx = px + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                float v[8];
                unpack8(*reinterpret_cast<const uint4*>(x + (((size_t)n * H + yy) * W + xx) * C + ch * 8), v);
                const int tap = (dy + 1) * 3 + (dx + 1);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = e * 9 + tap;                 // element index within the 72 (compile-time after unrolling)
                    const float wv = (idx & 1) ? bfhi(wr[idx >> 1]) : bflo(wr[idx >> 1]);
                    acc[e] += v[e] * wv;
                }
            }
        *reinterpret_cast<uint4*>(y + (size_t)pix * C + ch * 8) = pack8(acc);
    }
}

// y[n][c] = mean_p x[n][p][c]; one workgroup per (n, 64-channel slab): 256 threads = 32 pixel lanes x 8 chunks
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int HW, int C) {
    __shared__ float red[32][64];
    const int n = blockIdx.y, c0 = blockIdx.x * 64;
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < HW; p += 32) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p) * C + c0 + ch * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pl][ch * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int i = 0; i < 32; ++i) s += red[i][threadIdx.x];
        y[(size_t)n * C + c0 + threadIdx.x] = f2bf(s / (float)HW);
    }
}

// x[r][c] = act(x[r][c] + bias[c])  (bias may be null)
__global__ __launch_bounds__(256) void bias_act_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ bias, long rows, int C,
                                                       int act) {
    const int cpr = C >> 3;
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        float v[8], b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint4* p = reinterpret_cast<uint4*>(x + (i / cpr) * (long)C + ch * 8);
        unpack8(*p, v);
        if (bias) unpack8(*reinterpret_cast<const uint4*>(bias + ch * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_fn(bf2f(f2bf(v[e])) + b[e], act);
        *p = pack8(v);
    }
}

// x[n][p][c] *= gate[n][c]
__global__ __launch_bounds__(256) void scale_rows_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ gate, int N, int HW,
                                                         int C) {
    const int cpr = C >> 3;
    const long total = (long)N * HW * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const long pix = i / cpr;
        const long n = pix / HW;
        float v[8], gq[8];
        uint4* p = reinterpret_cast<uint4*>(x + pix * (long)C + ch * 8);
        unpack8(*p, v);
        unpack8(*reinterpret_cast<const uint4*>(gate + n * C + ch * 8), gq);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gq[e];
        *p = pack8(v);
    }
}

// x = act(x + y)
__global__ __launch_bounds__(256) void add_act_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ y, long n8, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float a[8], b[8];
        uint4* p = reinterpret_cast<uint4*>(x) + i;
        unpack8(*p, a);
        unpack8(reinterpret_cast<const uint4*>(y)[i], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = act_fn(a[e] + b[e], act);
        *p = pack8(a);
    }
}

// Conv3d k = s = 2, p = 1 gather: A[(to, yo, xo)][((kt*2+kh)*2+kw)*C + c] = x[2to-1+kt][2yo-1+kh][2xo-1+kw][c] (0 outside)
__global__ __launch_bounds__(256) void im2col3d_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ A, int T, int H, int W,
                                                       int C, int To, int Ho, int Wo) {
    const int cpr = C >> 3;
    const long total = (long)To * Ho * Wo * 8 * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const long rest = i / cpr;
        const int tap = (int)(rest & 7);
        const long row = rest >> 3;
        const int xo = (int)(row % Wo), yo = (int)((row / Wo) % Ho), to = (int)(row / ((long)Wo * Ho));
        const int t = 2 * to - 1 + (tap >> 2), yy = 2 * yo - 1 + ((tap >> 1) & 1), xx = 2 * xo - 1 + (tap & 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < T && yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = *reinterpret_cast<const uint4*>(x + (((size_t)t * H + yy) * W + xx) * C + ch * 8);
        *reinterpret_cast<uint4*>(A + ((size_t)row * 8 + tap) * C + ch * 8) = v;
    }
}

// Conv3d weight [Co][Ci][2][2][2] -> GEMM weight [Co][tap][Ci]
__global__ __launch_bounds__(256) void permute_conv3d_w_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Co, int Ci) {
    const long total = (long)Co * Ci * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int tap = (int)(i & 7);
        const long rest = i >> 3;
        const int ci = (int)(rest % Ci);
        const long co = rest / Ci;
        out[(co * 8 + tap) * Ci + ci] = w[i];
    }
}

int grid_for(long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }
}  // namespace

#define DONE return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP

int launch_dwconv3x3(const bf16_t* x, const bf16_t* w, bf16_t* y, int N, int H, int W, int C, hipStream_t s) {
    if (C % 8 || N < 1) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, s, x, w, y, N, H, W, C);
    DONE;
}
int launch_avgpool(const bf16_t* x, bf16_t* y, int N, int HW, int C, hipStream_t s) {
    if (C % 64 || N < 1) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(avgpool_kernel, dim3(C / 64, N), dim3(256), 0, s, x, y, HW, C);
    DONE;
}
int launch_bias_act(bf16_t* x, const bf16_t* bias, long rows, int C, int act, hipStream_t s) {
    if (C % 8 || rows < 1) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(bias_act_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, x, bias, rows, C, act);
    DONE;
}
int launch_scale_rows(bf16_t* x, const bf16_t* gate, int N, int HW, int C, hipStream_t s) {
    if (C % 8) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for((long)N * HW * (C / 8))), dim3(256), 0, s, x, gate, N, HW, C);
    DONE;
}
int launch_add_act(bf16_t* x, const bf16_t* y, long n, int act, hipStream_t s) {
    if (n % 8) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(add_act_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, x, y, n / 8, act);
    DONE;
}
int launch_im2col3d(const bf16_t* x, bf16_t* A, int T, int H, int W, int C, int To, int Ho, int Wo, hipStream_t s) {
    if (C % 8) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(im2col3d_kernel, dim3(grid_for((long)To * Ho * Wo * 8 * (C / 8))), dim3(256), 0, s, x, A, T, H, W, C, To, Ho, Wo);
    DONE;
}
int launch_permute_conv3d_w(const bf16_t* w, bf16_t* out, int Co, int Ci, hipStream_t s) {
    hipLaunchKernelGGL(permute_conv3d_w_kernel, dim3(grid_for((long)Co * Ci * 8)), dim3(256), 0, s, w, out, Co, Ci);
    DONE;
}
