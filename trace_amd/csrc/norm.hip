// LayerNorm (CLIP, eps 1e-5) and RMSNorm (Mistral, eps from config): HBM-bound row kernels.
// One 64-lane wave per row, the row is held in registers (16-byte loads, D <= 4096), statistics in fp32
// (two-pass mean/variance on the register copy), 16-byte stores.  4 rows per 256-thread block.
#include "common.h"
#include "kernels.h"

namespace {
constexpr int MAXCH = 8;   // 8 chunks x 64 lanes x 8 elements = 4096

template <bool RMS>
// (y and res carry no __restrict__: the STC block tail calls this with y == res — each lane loads its res chunk before it stores the same y chunk)
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* y, int ldy,
                                                   const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, int rows,
                                                   int D, float eps, int silu, const bf16_t* res = nullptr, int ldres = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    float v[MAXCH][8];
    uint4 uwv[MAXCH], ubv[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {      // weights first: their latency overlaps the row load + reductions
        const int c = lane + i * 64;
        if (c < nch) {
            uwv[i] = *reinterpret_cast<const uint4*>(w + c * 8);
            if (!RMS) ubv[i] = *reinterpret_cast<const uint4*>(b + c * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint4 u = *reinterpret_cast<const uint4*>(xr + c * 8);
            v[i][0] = bflo(u.x); v[i][1] = bfhi(u.x); v[i][2] = bflo(u.y); v[i][3] = bfhi(u.y);
            v[i][4] = bflo(u.z); v[i][5] = bfhi(u.z); v[i][6] = bflo(u.w); v[i][7] = bfhi(u.w);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += RMS ? v[i][e] * v[i][e] : v[i][e];
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)D + eps);
    } else {
        mean = s / (float)D;
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
        rstd = rsqrtf(q / (float)D + eps);
    }
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint4 uw = uwv[i];
            float o[8];
            const float ww[8] = {bflo(uw.x), bfhi(uw.x), bflo(uw.y), bfhi(uw.y), bflo(uw.z), bfhi(uw.z), bflo(uw.w), bfhi(uw.w)};
            if (RMS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[i][e] * rstd * ww[e];
            } else {
                const uint4 ub = ubv[i];
                const float bb[8] = {bflo(ub.x), bfhi(ub.x), bflo(ub.y), bfhi(ub.y), bflo(ub.z), bfhi(ub.z), bflo(ub.w), bfhi(ub.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
                if (res) {              // the RegNet block's tail in one pass (STC connector): act(LN(x) + shortcut), LN output rounded like a separate op
                    const uint4 ur = *reinterpret_cast<const uint4*>(res + (size_t)row * ldres + c * 8);
                    const float rr[8] = {bflo(ur.x), bfhi(ur.x), bflo(ur.y), bfhi(ur.y), bflo(ur.z), bfhi(ur.z), bflo(ur.w), bfhi(ur.w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float t = bf2f(f2bf(o[e])) + rr[e]; o[e] = silu ? t / (1.f + __expf(-t)) : t; }
                } else if (silu) {      // LayerNorm2d + SiLU of timm's ConvNormAct (STC connector): LN output rounded like a separate op
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float t = bf2f(f2bf(o[e])); o[e] = t / (1.f + __expf(-t)); }
                }
            }
            uint4 out;
            out.x = pack2bf(o[0], o[1]); out.y = pack2bf(o[2], o[3]); out.z = pack2bf(o[4], o[5]); out.w = pack2bf(o[6], o[7]);
            *reinterpret_cast<uint4*>(yr + c * 8) = out;
        }
    }
}

// D = 1024 (CLIP width) fast path: one wave normalises RW rows at once — all 2*RW 16-byte row loads are in flight before
// the first reduction (the one-row form keeps only two loads per lane outstanding and ran at 3.6 TB/s), non-temporal
// stores (the output is consumed by the next GEMM through L2/MALL anyway).
template <bool RMS, int RW>
__global__ __launch_bounds__(256) void norm1024_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                       const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, int rows,
                                                       float eps, int silu) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= rows) return;
    u32x4 xv[RW][2];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const int row = min(row0 + q, rows - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) xv[q][i] = *reinterpret_cast<const u32x4*>(x + (size_t)row * ldx + (lane + i * 64) * 8);
    }
    u32x4 wv[2], bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        wv[i] = *reinterpret_cast<const u32x4*>(w + (lane + i * 64) * 8);
        if (!RMS) bv[i] = *reinterpret_cast<const u32x4*>(b + (lane + i * 64) * 8);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        float v[2][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][2 * e] = bflo(xv[q][i][e]); v[i][2 * e + 1] = bfhi(xv[q][i][e]);
                s += RMS ? v[i][2 * e] * v[i][2 * e] + v[i][2 * e + 1] * v[i][2 * e + 1] : v[i][2 * e] + v[i][2 * e + 1];
            }
        s = wave_sum(s);
        float mean = 0.f, rstd;
        if (RMS) {
            rstd = rsqrtf(s / 1024.f + eps);
        } else {
            mean = s / 1024.f;
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; qq += d * d; }
            qq = wave_sum(qq);
            rstd = rsqrtf(qq / 1024.f + eps);
        }
        if (row0 + q >= rows) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w0 = bflo(wv[i][e]), w1 = bfhi(wv[i][e]);
                if (RMS) { o[2 * e] = v[i][2 * e] * rstd * w0; o[2 * e + 1] = v[i][2 * e + 1] * rstd * w1; }
                else {
                    o[2 * e] = (v[i][2 * e] - mean) * rstd * w0 + bflo(bv[i][e]);
                    o[2 * e + 1] = (v[i][2 * e + 1] - mean) * rstd * w1 + bfhi(bv[i][e]);
                }
            }
            if (!RMS && silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float t = bf2f(f2bf(o[e])); o[e] = t / (1.f + __expf(-t)); }
            }
            const u32x4 out = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
            __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(y + (size_t)(row0 + q) * ldy + (lane + i * 64) * 8));
        }
    }
}
}  // namespace

int launch_layernorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const bf16_t* w, const bf16_t* b, int rows, int D,
                     float eps, hipStream_t s, int silu, const bf16_t* res, int ldres) {
    if (D % 8 || D > MAXCH * 512 || (ldx % 8) || (ldy % 8) || rows <= 0 || (res && (ldres % 8))) return TRACE_ERR_ARG;
    if (res) {
        hipLaunchKernelGGL(norm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, D, eps, silu, res, ldres);
        return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
    }
    if (D == 1024 && rows >= 4096) {
        hipLaunchKernelGGL((norm1024_kernel<false, 4>), dim3((rows + 15) / 16), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, eps, silu);
        return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
    }
    hipLaunchKernelGGL(norm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, D, eps, silu, nullptr, 0);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_rmsnorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const bf16_t* w, int rows, int D, float eps,
                   hipStream_t s) {
    if (D % 8 || D > MAXCH * 512 || (ldx % 8) || (ldy % 8) || rows <= 0) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(norm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, y, ldy, w, nullptr, rows, D, eps, 0, nullptr, 0);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
