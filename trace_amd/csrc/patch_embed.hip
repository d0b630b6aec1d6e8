// CLIP ViT front end as ONE kernel: the stride-P patch convolution as an MFMA GEMM that reads the frame tensor directly — no im2col
// matrix — with the CLS / position-embedding assembly and `pre_layrnorm` in its epilogue.
// Reference: HF CLIPVisionEmbeddings (Conv2d 3 -> D, k = s = P, no bias; cat CLS; + position_embedding) and CLIPVisionTransformer.pre_layrnorm
// (transformers modeling_clip.py:148-154,208-217,641-642), reached from trace/model/multimodal_encoder/clip_encoder.py:50.
//
// Shape of the work (ViT-L/14-336: 576 patches per frame, K = 3 * 14 * 14 = 588, D = 1024): 0.69 GFLOP per frame — small next to a layer, but
// the round-3 path spent three passes on it (im2col 149 us + GEMM 128 us + assemble ~110 us + row statistics 35 us per 170 frames) because the
// A operand was materialised.  Here:
//   * K is re-indexed (c, ky, j) with j padded 14 -> 16, so a k-step of 32 is two (channel, patch-row) pairs and a lane's MFMA fragment
//     (8 consecutive k) is 8 CONSECUTIVE PIXELS of one frame row: one 16-byte load at 4-byte alignment straight from the frame tensor — every
//     pixel is read exactly once, from lines its 16-lane group uses completely.  The weights are repacked once at load into that k order, in
//     fragment order ([D/16][k-step][64 lanes][8]: a wave's weight fragment is 1 KB contiguous), zero in the two pad columns.
//   * a workgroup = 64 consecutive patches of one frame x all D channels: 8 waves, wave w holds channels [w D/8, (w+1) D/8) for the 64 patches
//     (4 x D/128 accumulator tiles); the 4 KB X tile of a k-step is loaded once (512 lanes, one fragment each) and shared through LDS,
//     double-buffered, one barrier per k-step; the weight fragments stream from L2 with a one-step register prefetch.
//   * epilogue, the round-3 arithmetic with the same rounding points: pe = bf16(acc); e = bf16(pe + pos[j]); two-pass LayerNorm in fp32 over the
//     D channels of a row (row sums: lane -> 4 lanes of a row by DPP-free xor shuffles -> 8 waves through LDS, fixed order); y = bf16(LN(e));
//     rows staged per wave in LDS and stored as 256-byte row segments.  The CLS row (identical for every frame) is made once at load and copied by the frame's first workgroup.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PE_MT = 4;                 // 16-row m-tiles per workgroup: 64 patches
constexpr int PE_WAVES = 8;
constexpr int PE_PITCH = 272;            // bytes per staged row of a wave's 128-channel slice (256 + 16: ds_write_b64 / ds_read_b128 conflict-free enough)

struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };

// 8 consecutive elements of a frame row starting at element offset `off` (4-byte aligned), as packed 16-bit pairs
template <typename TIN>
__device__ __forceinline__ uint4 load8(const TIN* base, size_t off) {
    if constexpr (sizeof(TIN) == 2) {
        const U4 v = *reinterpret_cast<const U4*>(reinterpret_cast<const bf16_t*>(base) + off);
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        const float* p = reinterpret_cast<const float*>(base) + off;
        const U4 a = *reinterpret_cast<const U4*>(p), b = *reinterpret_cast<const U4*>(p + 4);
        return make_uint4(pack2bf(__uint_as_float(a.x), __uint_as_float(a.y)), pack2bf(__uint_as_float(a.z), __uint_as_float(a.w)),
                          pack2bf(__uint_as_float(b.x), __uint_as_float(b.y)), pack2bf(__uint_as_float(b.z), __uint_as_float(b.w)));
    }
}

// NTW = D / 128 accumulator n-tiles per wave
template <typename TIN, int NTW>
__global__ __launch_bounds__(512) void patch_embed_kernel(const TIN* __restrict__ frames, const bf16_t* __restrict__ wp, const bf16_t* __restrict__ pos,
                                                          const bf16_t* __restrict__ lw, const bf16_t* __restrict__ lb, const bf16_t* __restrict__ cls_row,
                                                          bf16_t* __restrict__ X, int S, int P, int G, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = reinterpret_cast<uint4*>(smem);                                   // [2][PE_MT * 64] X fragments of a k-step
    float* red = reinterpret_cast<float*>(smem + 2 * PE_MT * 1024);               // [PE_WAVES][64] row partials
    unsigned char* stage = smem + 2 * PE_MT * 1024 + PE_WAVES * 64 * 4;           // [PE_WAVES][64 rows][PE_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 15, g = lane >> 4;
    const int GG = G * G, NT = GG + 1, KS = (3 * P * 16) / 32;
    const int wg_per_frame = (GG + 63) / 64;
    const int t = blockIdx.x / wg_per_frame, m0 = (blockIdx.x - t * wg_per_frame) * 64;

    // ---- loader role of this thread (tid < PE_MT * 64): the fragment of lane (lr, lg) of m-tile lmt ----
    const bool ldr = tid < PE_MT * 64;
    const int lmt = tid >> 6;
    const int lpatch = min(m0 + lmt * 16 + r, GG - 1);          // rows past the frame's patches repeat the last one (never stored)
    const int lgy = lpatch / G, lgx = lpatch - lgy * G;
    const int j0 = (g & 1) * 8;
    // the window [px0, px0 + 8) of the half with j0 = 8 runs past the frame row on the last patch column: read 8 - ov pixels earlier, shift
    const int ov = max(0, lgx * P + j0 + 8 - S);                 // 0 or 2 for P = 14 (P = 16: always 0)
    auto xload = [&](int ks) -> uint4 {
        const int pk = ks * 2 + (g >> 1);                        // (channel, patch row) pair of this half k-step
        const int c = pk / P, ky = pk - c * P;
        const size_t off = (((size_t)t * 3 + c) * S + (size_t)(lgy * P + ky)) * S + lgx * P + j0 - ov;
        uint4 v = load8<TIN>(frames, off);
        if (ov) { v.x = v.y; v.y = v.z; v.z = v.w; v.w = 0u; }  // ov == 2 elements == one dword
        if (j0 && P < 16) {                                       // the pad columns j >= P carry zero weights; zero the pixels too (a NaN pixel stays in its own patch)
            if (P <= 14) v.w = 0u;
            if (P <= 12) v.z = 0u;
            if (P <= 10) v.y = 0u;
        }
        return v;
    };
    // ---- this wave's weight fragments: n-tiles wid * NTW .. + NTW, packed [ntile][KS][64][8] ----
    const bf16_t* wbase = wp + ((size_t)(wid * NTW) * KS * 64 + lane) * 8;
    bf16x8_t wf[NTW], wn[NTW];
    auto wload = [&](bf16x8_t (&dst)[NTW], int ks) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) dst[j] = *reinterpret_cast<const bf16x8_t*>(wbase + ((size_t)j * KS + ks) * 512);
    };
    f32x4_t acc[PE_MT][NTW];
#pragma unroll
    for (int i = 0; i < PE_MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 xr = make_uint4(0u, 0u, 0u, 0u);
    if (ldr) xr = xload(0);
    wload(wf, 0);
    for (int ks = 0; ks < KS; ++ks) {
        if (ldr) xs[(ks & 1) * PE_MT * 64 + tid] = xr;
        __syncthreads();
        if (ks + 1 < KS) {
            if (ldr) xr = xload(ks + 1);
            wload(wn, ks + 1);
        }
#pragma unroll
        for (int i = 0; i < PE_MT; ++i) {
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, xs[(ks & 1) * PE_MT * 64 + i * 64 + lane]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = mfma16(wf[j], a, acc[i][j]);
        }
        if (ks + 1 < KS) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) wf[j] = wn[j];
        }
    }

    // ---- epilogue: lane (r, g) holds, for m-tile i and n-tile j, channels n = (wid * NTW + j) * 16 + 4 g + e of patch row m0 + 16 i + r ----
    const int nbase = wid * NTW * 16 + 4 * g;
    // a row's sum over the workgroup: the lane's values -> its 4 g-lanes (xor 16, 32) -> 8 waves through LDS, in wave order
    auto row_sum = [&](float (&v)[PE_MT]) {
#pragma unroll
        for (int i = 0; i < PE_MT; ++i) {
            v[i] += __shfl_xor(v[i], 16, 64);
            v[i] += __shfl_xor(v[i], 32, 64);
        }
        __syncthreads();                                   // `red` is free again
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < PE_MT; ++i) red[wid * 64 + i * 16 + r] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PE_MT; ++i) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < PE_WAVES; ++w) s += red[w * 64 + i * 16 + r];
            v[i] = s;
        }
    };
    // e = bf16(bf16(acc) + pos): kept in the accumulators as fp32 values of bf16 numbers
    float s1[PE_MT];
#pragma unroll
    for (int i = 0; i < PE_MT; ++i) {
        const int patch = min(m0 + i * 16 + r, GG - 1);
        const bf16_t* prow = pos + (size_t)(1 + patch) * D + nbase;
        s1[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const uint2 pv = *reinterpret_cast<const uint2*>(prow + j * 16);
            f32x4_t a = acc[i][j];
            a[0] = bf2f(f2bf(bf2f(f2bf(a[0])) + bflo(pv.x)));
            a[1] = bf2f(f2bf(bf2f(f2bf(a[1])) + bfhi(pv.x)));
            a[2] = bf2f(f2bf(bf2f(f2bf(a[2])) + bflo(pv.y)));
            a[3] = bf2f(f2bf(bf2f(f2bf(a[3])) + bfhi(pv.y)));
            acc[i][j] = a;
            s1[i] += (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    row_sum(s1);
    float mean[PE_MT], q[PE_MT];
#pragma unroll
    for (int i = 0; i < PE_MT; ++i) {
        mean[i] = s1[i] / (float)D;
        q[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[i][j][e] - mean[i]; q[i] = fmaf(d, d, q[i]); }
    }
    row_sum(q);
    // y = bf16((e - mean) * rstd * gamma + beta), staged in LDS
    unsigned char* st = stage + (size_t)wid * 64 * PE_PITCH;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const uint2 gw = *reinterpret_cast<const uint2*>(lw + nbase + j * 16), gb = *reinterpret_cast<const uint2*>(lb + nbase + j * 16);
        const float w4[4] = {bflo(gw.x), bfhi(gw.x), bflo(gw.y), bfhi(gw.y)}, b4[4] = {bflo(gb.x), bfhi(gb.x), bflo(gb.y), bfhi(gb.y)};
#pragma unroll
        for (int i = 0; i < PE_MT; ++i) {
            const float rstd = rsqrtf(q[i] / (float)D + eps);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (acc[i][j][e] - mean[i]) * rstd * w4[e] + b4[e];
            const uint2 o = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
            *reinterpret_cast<uint2*>(st + (i * 16 + r) * PE_PITCH + (j * 16 + 4 * g) * 2) = o;
        }
    }
    __syncthreads();                                        // the staged rows of every wave are complete (and `red` reads are done)
    // store: the wave's 64 rows x (NTW * 16 channels = NTW * 32 bytes): lanes cover a row segment with 16-byte pieces
    constexpr int PIECES = NTW * 2;                         // 16-byte pieces per row segment (NTW = 8: 16 pieces = 256 bytes)
    constexpr int ROWS_PER_PASS = 64 / PIECES;
#pragma unroll 4
    for (int rr = 0; rr < 64; rr += ROWS_PER_PASS) {
        const int row = rr + lane / PIECES, piece = lane % PIECES;
        const int patch = m0 + row;
        if (patch < GG) {
            const uint4 v = *reinterpret_cast<const uint4*>(st + row * PE_PITCH + piece * 16);
            *reinterpret_cast<uint4*>(X + ((size_t)t * NT + 1 + patch) * D + wid * NTW * 16 + piece * 8) = v;
        }
    }
    // the frame's CLS row (made once at load: LN(bf16(cls + pos[0])))
    if (m0 == 0) {
        for (int c = tid; c < (D >> 3); c += 512)
            *reinterpret_cast<uint4*>(X + (size_t)t * NT * D + c * 8) = *reinterpret_cast<const uint4*>(cls_row + c * 8);
    }
}

// W [D][ldw] (k = c P P + ky P + j, the checkpoint's conv weight flattened) -> wp [D/16][KS][64 lanes][8] in the kernel's k order
// (kk = (c P + ky) 16 + j, zero for j >= P), lane (r, g) of (ntile, ks) holding W[ntile 16 + r][ks 32 + g 8 .. + 8)
__global__ __launch_bounds__(256) void patch_pack_kernel(const bf16_t* __restrict__ W, int ldw, bf16_t* __restrict__ wp, int D, int P) {
    const int KS = (3 * P * 16) / 32;
    const long total = (long)(D / 16) * KS * 64;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const long rest = idx >> 6;
        const int ks = (int)(rest % KS), ntile = (int)(rest / KS);
        const int n = ntile * 16 + (lane & 15), g = lane >> 4;
        uint32_t o[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            bf16_t h[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int kk = ks * 32 + g * 8 + e2 * 2 + q, pk = kk >> 4, j = kk & 15;
                const int c = pk / P, ky = pk - c * P;
                h[q] = j < P ? W[(size_t)n * ldw + (c * P + ky) * P + j] : (bf16_t)0;
            }
            o[e2] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
        }
        *reinterpret_cast<uint4*>(wp + idx * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// cls_row = bf16(LN(bf16(cls + pos[0]))) (two-pass, fp32).  One workgroup of 256 threads, D <= 1024.
__global__ __launch_bounds__(256) void cls_row_kernel(const bf16_t* __restrict__ cls, const bf16_t* __restrict__ pos, const bf16_t* __restrict__ lw,
                                                      const bf16_t* __restrict__ lb, bf16_t* __restrict__ out, int D, float eps) {
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) s_red[wid] = v;
        __syncthreads();
        return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    };
    float e[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        e[i] = c < D ? bf2f(f2bf(bf2f(cls[c]) + bf2f(pos[c]))) : 0.f;
        s += e[i];
    }
    const float mean = block_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (tid + i * 256 < D) { const float d = e[i] - mean; q += d * d; }
    const float rstd = rsqrtf(block_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        if (c < D) out[c] = f2bf((e[i] - mean) * rstd * bf2f(lw[c]) + bf2f(lb[c]));
    }
}

template <typename TIN, int NTW>
int launch_pe(const void* frames, const bf16_t* wp, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb, const bf16_t* cls_row,
              bf16_t* X, int T, int S, int P, int G, int D, float eps, hipStream_t s) {
    constexpr int LDS = 2 * PE_MT * 1024 + PE_WAVES * 64 * 4 + PE_WAVES * 64 * PE_PITCH;
    static LdsGrant grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(patch_embed_kernel<TIN, NTW>), LDS)) return TRACE_ERR_HIP;
    const int grid = T * ((G * G + 63) / 64);
    hipLaunchKernelGGL((patch_embed_kernel<TIN, NTW>), dim3(grid), dim3(512), LDS, s, (const TIN*)frames, wp, pos, lw, lb, cls_row, X, S, P,
                       G, D, eps);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

}  // namespace

bool patch_embed_supported(int S, int P, int D) { return (P == 14 || P == 16) && S % P == 0 && (D == 128 || D == 256 || D == 512 || D == 1024); }
size_t patch_embed_packed_elems(int P, int D) { return (size_t)D * 3 * P * 16; }

int launch_patch_pack(const bf16_t* W, int ldw, bf16_t* wp, int D, int P, hipStream_t s) {
    if (!patch_embed_supported(P, P, D)) return TRACE_ERR_ARG;
    const long total = (long)(D / 16) * ((3 * P * 16) / 32) * 64;
    hipLaunchKernelGGL(patch_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, ldw, wp, D, P);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_cls_row(const bf16_t* cls, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb, bf16_t* out, int D, float eps, hipStream_t s) {
    if (D < 8 || D > 1024) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(cls_row_kernel, dim3(1), dim3(256), 0, s, cls, pos, lw, lb, out, D, eps);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

// frames [T, 3, S, S] (16-bit elements, or fp32 when frames_fp32) -> X [T, G G + 1, D] = pre_layrnorm(cat(CLS, conv(frames)) + pos)
int launch_patch_embed(const void* frames, int frames_fp32, const bf16_t* wp, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb,
                       const bf16_t* cls_row, bf16_t* X, int T, int S, int P, int D, float eps, hipStream_t s) {
    if (T < 1 || !patch_embed_supported(S, P, D)) return TRACE_ERR_ARG;
    const int G = S / P;
#define PE_GO(NTW_)                                                                                                                                  \
    return frames_fp32 ? launch_pe<float, NTW_>(frames, wp, pos, lw, lb, cls_row, X, T, S, P, G, D, eps, s)                                          \
                       : launch_pe<bf16_t, NTW_>(frames, wp, pos, lw, lb, cls_row, X, T, S, P, G, D, eps, s)
    switch (D / 128) {
        case 1: PE_GO(1);
        case 2: PE_GO(2);
        case 4: PE_GO(4);
        default: PE_GO(8);
    }
#undef PE_GO
}
