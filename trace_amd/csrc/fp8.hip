// fp8 (OCP e4m3fn) weight path of the Mistral decoder — BASELINE config 5 ("TRACE-uni long video, 256 frames, fp8 MFMA weight
// path").  The reference has no fp8 mode (its only quantised loading is bitsandbytes, trace/model/builder.py:39-48), so this is
// this build's own W8A8 scheme and its parity anchor is the bf16 path (tests/test_gpu_fp8.py):
//   * weights: one fp32 scale per OUTPUT ROW (amax / 448), quantised once at load from the bf16 copy;
//   * activations: one fp32 scale per TOKEN ROW, quantised on the fly by quant_rows_fp8 (dynamic, symmetric);
//   * both operands go to the fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate), the two scales are applied to the
//     accumulators in the epilogue.  Norm weights, embeddings, heads, KV cache and the ViT stay bf16.
// What it buys: the decode step streams 7 GB of layer weights instead of 14 (the step is HBM-bound), and the prefill GEMMs move
// half the operand bytes through the LDS-DMA path at the same MFMA rate (the non-scaled fp8 MFMA runs at the bf16 rate).
// This file: row quantiser, the decode tile layout for fp8 and the fp8 decode GEMV; the fp8 GEMM is gemm.hip / gemm_ldr.hip's
// kernel instantiated with FP8 = true.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr float FP8_MAX = 448.f;

__device__ __forceinline__ uint32_t cvt4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

// X [rows][K] bf16 (row stride ldx) -> X8 [rows][K] e4m3 bytes (row stride ld8) + sx[row] = amax / 448 (1 for an all-zero row).
// q = rne(x * (448 / amax)), |x * inv| <= 448 by construction (clamped against the last-ulp case).  One workgroup per row; the
// row stays in registers between the amax pass and the conversion (K <= 16384).
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ X, long ldx, uint8_t* __restrict__ X8, long ld8,
                                                             float* __restrict__ sx, int K) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = X + (size_t)row * ldx;
    constexpr int MAXC = 8;
    uint4 v[MAXC];
    float amax = 0.f;
    const int nch = K >> 3;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + i * 256;
        v[i] = make_uint4(0, 0, 0, 0);
        if (c < nch) {
            v[i] = *reinterpret_cast<const uint4*>(xr + (size_t)c * 8);
            const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bflo(u[e])), fabsf(bfhi(u[e]))));
        }
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) s_red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float scale = amax > 0.f ? amax / FP8_MAX : 1.f;
    const float inv = amax > 0.f ? FP8_MAX / amax : 1.f;
    if (tid == 0) sx[row] = scale;
    uint8_t* qr = X8 + (size_t)row * ld8;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + i * 256;
        if (c < nch) {
            const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = fminf(fmaxf(bflo(u[e]) * inv, -FP8_MAX), FP8_MAX);
                f[2 * e + 1] = fminf(fmaxf(bfhi(u[e]) * inv, -FP8_MAX), FP8_MAX);
            }
            *reinterpret_cast<uint2*>(qr + (size_t)c * 8) = make_uint2(cvt4_fp8(f[0], f[1], f[2], f[3]), cvt4_fp8(f[4], f[5], f[6], f[7]));
        }
    }
}

// Row-major W8 [N][K] bytes -> decode copy [N/16][K/128][64 lanes][32 B]: lane (r = lane & 15, g = lane >> 4) of block (tile, unit)
// holds W8[tile*16 + r][unit*128 + g*32 .. +32] = its four 8-byte MFMA fragments back to back.  One thread per 16-byte piece.
__global__ __launch_bounds__(256) void tile_pack_fp8_kernel(const uint8_t* __restrict__ src, long ldw, uint8_t* __restrict__ dst, int N, int K) {
    const int U = K >> 7;
    const long total = (long)N * (K >> 4);
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int n = (int)(p / (K >> 4)), k16 = (int)(p - (long)n * (K >> 4));
        const int tile = n >> 4, r = n & 15, u = k16 >> 3, g = (k16 & 7) >> 1, hh = k16 & 1;
        *reinterpret_cast<u32x4_t*>(dst + (((size_t)tile * U + u) * 64 + g * 16 + r) * 32 + hh * 16) =
            *reinterpret_cast<const u32x4_t*>(src + (size_t)n * ldw + (size_t)k16 * 16);
    }
}

__device__ __forceinline__ uint4 ldg_nt16(const uint8_t* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ long lo64(const uint4& u) { return (long)(((unsigned long long)u.y << 32) | u.x); }
__device__ __forceinline__ long hi64(const uint4& u) { return (long)(((unsigned long long)u.w << 32) | u.z); }

// Decode GEMV on fp8: the structure of decode.hip's skinny_lds_kernel (activations of a K-chunk parked in LDS in fragment order,
// weights streamed once from the tile-contiguous copy straight into the MFMA A operand, fp32 k-chunk partial rows out) with
// 128-k units: a lane's 32 contiguous weight bytes are FOUR fragments, and a chunk that fits 128 KB of LDS is twice as many k wide,
// so K is cut into half as many chunks (half the partial-row traffic).  partial[ks][m][n] = acc * sx[m] * sw[n].
// WONLY (round 3, the "weight-only" scheme, W8A16): the same e4m3 weight stream, but the activations stay bf16 — X8 then points at bf16 rows
// (ldx in elements), a 128-k unit parks FOUR 1 KB images per row group (the lane's 32 k as 4 x 8 bf16) instead of two, and each 8-byte
// weight fragment is widened to bf16 in registers (v_cvt_scalef32_pk_bf16_fp8, exact: every e4m3 value is a bf16 value) for the bf16
// MFMA.  Half the quantisation noise of W8A8 in variance (only one operand is rounded), the same weight bytes; a chunk that fits LDS is
// half as many k wide again (twice the partial rows of the W8A8 GEMV, as many as the bf16 GEMV).  partial[ks][m][n] = acc * sw[n].
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16x8_t widen8(long w) {
    const unsigned int d0 = (unsigned int)((unsigned long long)w & 0xffffffffull), d1 = (unsigned int)((unsigned long long)w >> 32);
    union { bf16x8_t v; bf16x2_t p[4]; } u;
    u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, false);
    u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, true);
    u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, false);
    u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, true);
    return u.v;
}
template <int NB, int NT, bool WONLY = false>
__global__ __launch_bounds__(512) void skinny_fp8_kernel(const uint8_t* __restrict__ X8, long ldx, const float* __restrict__ sx,
                                                         const uint8_t* __restrict__ W, const float* __restrict__ sw, int B, int K,
                                                         int chunk_units, int KS, int T, int WPT, int ntiles, float* __restrict__ ws, int N) {
    constexpr int UN = 2;                                // 128-k units per load batch: 4 KB (NT = 1) / 8 KB (NT = 2) of weights per wave
    constexpr int NF = NT * NB;
    constexpr int IPU = WONLY ? 4 : 2;                   // 1 KB activation images per unit and row group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);                                       // [unit][half / quarter][nb][lane]
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + (size_t)chunk_units * IPU * NB * 1024);  // [task][wsub][NF][lane]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int ks = blockIdx.x % KS, rg = blockIdx.x / KS;
    const int U = K >> 7;
    const int u_beg = ks * chunk_units, nu = min(U - u_beg, chunk_units);
    const int team = wid / WPT, wsub = wid - team * WPT, nteams = nwaves / WPT;
    const int ua = (wsub * nu) / WPT, ub = ((wsub + 1) * nu) / WPT;
    const uint8_t* wp[NT];
    uint4 wa[UN][NT][2], wb[UN][NT][2];
    auto loadw = [&](uint4 (&wf)[UN][NT][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const bool ok = u + j < ub;
            const size_t ko = (size_t)(u + j) * 2048;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wf[j][t][0] = ok ? ldg_nt16(wp[t] + ko) : make_uint4(0, 0, 0, 0);
                wf[j][t][1] = ok ? ldg_nt16(wp[t] + ko + 16) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    f32x4_t acc[NT][NB];
    bool parked = false;
    for (int task = team; task < T; task += nteams) {
    const int tile = rg * T + task;
    const bool active = tile < ntiles;
    const int n0 = tile * 16 * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + ((size_t)(active ? tile * NT + t : 0) * U + u_beg) * 2048 + lane * 32;
    const bool work = active && ua < ub;
    if (work) loadw(wa, ua);
    // park X8[:, chunk]: combo c = (unit*2 + half)*NB + nb, lane (r, g) holds X8[16 nb + r][(u_beg + unit)*128 + g*32 + half*16 .. +16]
    if (!parked) {                                       // by LDS-DMA, one lane-linear 1 KB image per combo (see decode.hip); rows >= B read row B-1
        const int combos = nu * IPU * NB;
        for (int c = wid; c < combos; c += nwaves) {
            const int nb = c % NB, uh = c / NB, hh = uh % IPU, u = uh / IPU;
            const int m = min(16 * nb + r, B - 1);
            // W8A8: 16 e4m3 bytes of the lane's 32-k group; weight-only: 8 bf16 (quarter hh) of it
            const uint8_t* src = WONLY ? X8 + ((size_t)m * ldx + (size_t)(u_beg + u) * 128 + g * 32 + hh * 8) * 2
                                       : X8 + (size_t)m * ldx + (size_t)(u_beg + u) * 128 + g * 32 + hh * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + (size_t)c * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        parked = true;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](uint4 (&wf)[UN][NT][2], int u) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            if (WONLY) {
                if (u + j < ub) {
                    bf16x8_t wq[NT][4];                  // the lane's 32 weights per tile, widened once, used by every row group
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        wq[t][0] = widen8(lo64(wf[j][t][0])); wq[t][1] = widen8(hi64(wf[j][t][0]));
                        wq[t][2] = widen8(lo64(wf[j][t][1])); wq[t][3] = widen8(hi64(wf[j][t][1]));
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bf16x8_t xq = __builtin_bit_cast(bf16x8_t, xs[(((u + j) * 4 + q) * NB + nb) * 64 + lane]);
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][q], xq, acc[t][nb], 0, 0, 0);
                        }
                }
                continue;
            }
            if (u + j < ub) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const u32x4_t a0 = xs[(((u + j) * 2 + 0) * NB + nb) * 64 + lane], a1 = xs[(((u + j) * 2 + 1) * NB + nb) * 64 + lane];
                    const uint4 x0 = make_uint4(a0[0], a0[1], a0[2], a0[3]), x1 = make_uint4(a1[0], a1[1], a1[2], a1[3]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lo64(wf[j][t][0]), lo64(x0), acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(hi64(wf[j][t][0]), hi64(x0), acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lo64(wf[j][t][1]), lo64(x1), acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(hi64(wf[j][t][1]), hi64(x1), acc[t][nb], 0, 0, 0);
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
    if (WPT > 1) {                                       // the WPT waves of a task: fixed-order sum through LDS (workgroup-uniform branch)
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
    if (!active || wsub != 0) continue;
    float* pr = ws + (size_t)ks * SK_ROWS * N;           // lane (r, g): out[m = 16 nb + r][n0 + t*16 + g*4 + 0..3]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4_t wsc = *reinterpret_cast<const f32x4_t*>(sw + n0 + t * 16 + g * 4);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int m = 16 * nb + r;
            if (m < B) {
                const float a = WONLY ? 1.f : sx[m];
                *reinterpret_cast<f32x4_t*>(pr + (size_t)m * N + n0 + t * 16 + g * 4) =
                    f32x4_t{acc[t][nb][0] * a * wsc[0], acc[t][nb][1] * a * wsc[1], acc[t][nb][2] * a * wsc[2], acc[t][nb][3] * a * wsc[3]};
            }
        }
    }
    }   // task loop
}

struct Fp8Plan { int KS, chunk_units, T, WPT, ntiles, grid, threads, NT, NB; };
int fp8_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
// same partition rule as decode.hip's skinny_plan, in 128-k units (a unit is the same 2 KB of weights per 16 rows)
Fp8Plan fp8_plan(int N, int K, int B, bool wonly = false) {
    Fp8Plan p{};
    p.NT = (N >= 16384 && N % 32 == 0) ? 2 : 1;
    p.NB = B > 32 ? 4 : B > 16 ? 2 : 1;
    const int U = K / 128;
    const int cap = (wonly ? 32 : 64) / p.NB;            // 128-k units of activations that fit 128 KB of LDS (bf16 rows: half as many)
    const int ks_min = (U + cap - 1) / cap, ks_max = std::min(U, ks_min + 4);
    p.ntiles = N / (16 * p.NT);
    const int ncu = fp8_num_cus();
    long best = -1;
    for (int ks = ks_min; ks <= ks_max; ++ks) {
        const int chunk = (U + ks - 1) / ks;
        for (int T = 1; T <= 16; ++T) {
            const int grid = ks * ((p.ntiles + T - 1) / T);
            const long rounds = (grid + ncu - 1) / ncu;
            const long cost = rounds * T * chunk * 64 + rounds * 8 + ks;
            if (best < 0 || cost < best) { best = cost; p.KS = ks; p.chunk_units = chunk; p.T = T; p.grid = grid; }
        }
    }
    p.WPT = 1;
    while (p.WPT * 2 * p.T <= 8 && p.WPT * 2 * p.T * p.NT * p.NB <= 28 && p.WPT * 2 <= p.chunk_units) p.WPT *= 2;
    p.threads = std::min(p.T, 8) * p.WPT * 64;
    return p;
}

template <int NB, int NT, bool WONLY = false>
int fp8_launch(const Fp8Plan& p, const uint8_t* X8, long ldx, const float* sx, const uint8_t* W, const float* sw, int B, int N, int K, float* ws,
               hipStream_t s) {
    const size_t lds = (size_t)p.chunk_units * (WONLY ? 4 : 2) * NB * 1024 + (p.WPT > 1 ? (size_t)p.T * p.WPT * NT * NB * 1024 : 0);
    static LdsGrantSized grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(skinny_fp8_kernel<NB, NT, WONLY>), lds)) return TRACE_ERR_HIP;
    hipLaunchKernelGGL((skinny_fp8_kernel<NB, NT, WONLY>), dim3(p.grid), dim3(p.threads), lds, s, X8, ldx, sx, W, sw, B, K, p.chunk_units, p.KS, p.T,
                       p.WPT, p.ntiles, ws, N);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
}  // namespace

int launch_quant_rows_fp8(const bf16_t* X, long ldx, uint8_t* X8, long ld8, float* sx, int rows, int K, hipStream_t s) {
    if (rows < 1 || K < 8 || K % 8 || K > 16384 || (ldx % 8) || (ld8 % 8)) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3(rows), dim3(256), 0, s, X, ldx, X8, ld8, sx, K);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_tile_pack_fp8(const uint8_t* src, long ldw, uint8_t* dst, int N, int K, hipStream_t s) {
    if (N % 16 || K % 128 || ldw % 16) return TRACE_ERR_ARG;
    const long total = (long)N * (K / 16);
    hipLaunchKernelGGL(tile_pack_fp8_kernel, dim3((int)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, s, src, ldw, dst, N, K);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int skinny_fp8_ks(int N, int K, int B) { return fp8_plan(N, K, B).KS; }
int skinny_w8_ks(int N, int K, int B) { return fp8_plan(N, K, B, true).KS; }

// the weight-only form: X bf16 [B][K] (ldx elements), e4m3 tile-layout weights + row scales; partial rows [skinny_w8_ks()][SK_ROWS][N]
int launch_skinny_w8(const bf16_t* X, long ldx, const uint8_t* Wtiled, const float* sw, int B, int N, int K, float* ws, size_t ws_floats,
                     hipStream_t s) {
    if (B < 1 || B > SKINNY_ROWS || K % 128 || N % 16 || (ldx % 8)) return TRACE_ERR_ARG;
    const Fp8Plan p = fp8_plan(N, K, B, true);
    if (!ws || ws_floats < (size_t)p.KS * SK_ROWS * N) return TRACE_ERR_ARG;
    const uint8_t* X8 = reinterpret_cast<const uint8_t*>(X);
#define WL(NT_) (B <= 16 ? fp8_launch<1, NT_, true>(p, X8, ldx, nullptr, Wtiled, sw, B, N, K, ws, s) \
               : B <= 32 ? fp8_launch<2, NT_, true>(p, X8, ldx, nullptr, Wtiled, sw, B, N, K, ws, s) \
                         : fp8_launch<4, NT_, true>(p, X8, ldx, nullptr, Wtiled, sw, B, N, K, ws, s))
    return p.NT == 2 ? WL(2) : WL(1);
#undef WL
}

// partial rows [skinny_fp8_ks()][SK_ROWS][N] fp32 in ws (>= KS * SK_ROWS * N floats): the consumers of decode.hip sum them
int launch_skinny_fp8(const uint8_t* X8, long ldx, const float* sx, const uint8_t* Wtiled, const float* sw, int B, int N, int K, float* ws,
                      size_t ws_floats, hipStream_t s) {
    if (B < 1 || B > SKINNY_ROWS || K % 128 || N % 16 || (ldx % 16)) return TRACE_ERR_ARG;
    const Fp8Plan p = fp8_plan(N, K, B);
    if (!ws || ws_floats < (size_t)p.KS * SK_ROWS * N) return TRACE_ERR_ARG;
#define FL(NT_) (B <= 16 ? fp8_launch<1, NT_>(p, X8, ldx, sx, Wtiled, sw, B, N, K, ws, s) \
               : B <= 32 ? fp8_launch<2, NT_>(p, X8, ldx, sx, Wtiled, sw, B, N, K, ws, s) \
                         : fp8_launch<4, NT_>(p, X8, ldx, sx, Wtiled, sw, B, N, K, ws, s))
    return p.NT == 2 ? FL(2) : FL(1);
#undef FL
}
