// Flash-style fused attention for the two prefill-shaped attentions on the path:
//   * CLIP ViT encoder attention: head_dim 64, 577 tokens/frame, non-causal (HF CLIPAttention; softmax fp32)
//   * Mistral prefill attention:  head_dim 128, causal, GQA 4 q-heads per kv-head (HF MistralAttention)
//
// CDNA4 mapping.  One wave owns 32 query rows and walks the keys in tiles of 64.  Both matmuls are issued
// "swapped" on the 32x32x16 bf16 MFMA so that every per-row quantity lives in the lane that owns the row:
//   S^T[kv,q] = K[kv,:] . Q[q,:]      (A = K fragment from LDS, B = Q fragment held in registers)
//   O^T[d,q] += V^T[d,kv] . P[q,kv]   (A = V^T fragment from LDS, B = P straight from the S^T accumulators)
// The C layout of the first MFMA (lane -> column q = lane&31, 16 kv rows) is, register for register, a legal
// B operand of the second one once the V^T fragment is read with the matching kv permutation, so P never
// leaves registers and the online-softmax state (m, l) and the O^T accumulators of a query row all sit in the
// same lane (row max/sum need one cross-half shuffle).  V arrives pre-transposed (V^T[d, kv], kv padded with
// zeros to a multiple of 64) from transpose_v below, so both LDS operands are K-contiguous 8/16-byte reads:
// K rows XOR-swizzled (conflict-free ds_read_b128), V^T rows padded to 136 B (conflict-free ds_read_b64).
// K/V^T tiles are double-buffered in LDS; the next tile's global loads are issued before the MFMAs of the
// current one and written to LDS after them (one barrier per tile).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BKV = 64;
constexpr int VROW = 136;   // bytes per V^T row in LDS: 64 kv * 2 + 8 pad

template <int HD>
__device__ __forceinline__ int kswz(int row, int kc) {
    return row * (HD * 2) + ((kc ^ (HD == 64 ? ((row >> 1) & 7) : (row & 15))) << 4);
}

template <int HD, bool GQA>
__device__ __forceinline__ void attn_body(AttnArgs a) {
    constexpr int KT_BYTES = BKV * HD * 2;
    constexpr int VT_BYTES = HD * VROW;
    constexpr int BUF = KT_BYTES + VT_BYTES;
    constexpr int NCH = BKV * HD / 8 / 256;          // 16-byte chunks per thread per operand tile
    constexpr int CPR = HD / 8;                      // chunks per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c32 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, kvh = blockIdx.y;
    const int qh = GQA ? kvh * (a.heads / a.kv_heads) + wid : kvh;
    const int qt = GQA ? blockIdx.x : blockIdx.x * 4 + wid;
    const int q0 = qt * 32;
    const bool active = q0 < a.nq_rows;
    const int off = a.nkv_rows - a.nq_rows;
    const int qabs = q0 + c32;

    // block-level kv extent
    int kv_limit = a.nkv_rows;
    if (a.causal) {
        const int qmax = (GQA ? q0 : blockIdx.x * 128 + 96) + 31 + off;   // last query row of the block
        kv_limit = min(kv_limit, qmax + 1);
    }
    const int nt = (kv_limit + BKV - 1) / BKV;

    // Q fragments (B operand): lane (q = c32, h) holds d = s*16 + h*8 .. +8
    bf16x8_t qf[HD / 16];
    {
        const int qr = min(qabs, a.nq_rows - 1);
        const bf16_t* qp = a.Q + (size_t)b * a.q_bs + (size_t)qh * a.q_hs + (size_t)qr * a.q_rs + h * 8;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16);
    }

    const bf16_t* kbase = a.K + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
    const bf16_t* vbase = a.V + (size_t)b * a.v_bs + (size_t)kvh * a.v_hs;
    int krow[NCH], kkc[NCH], vrow[NCH], vkc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * 256;
        krow[i] = c / CPR; kkc[i] = c % CPR;
        vrow[i] = c >> 3; vkc[i] = c & 7;
    }
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    u32x4 rk[NCH], rv[NCH];
    // (macros, not lambdas: with by-reference lambda captures hipcc kept rk/rv in scratch memory and the prefetch
    //  registers were spilled right after the loads, serialising the global-load latency)
#define ATTN_GLOAD(T_)                                                                                         \
    {                                                                                                          \
        const int kv0_ = (T_) * BKV;                                                                           \
        _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                                      \
            const int kr_ = min(kv0_ + krow[i], a.nkv_rows - 1);                                               \
            rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)kr_ * a.k_rs + kkc[i] * 8);                \
            rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)vrow[i] * a.v_rs + kv0_ + vkc[i] * 8);     \
        }                                                                                                      \
    }
#define ATTN_LSTORE(BUF_)                                                                                      \
    {                                                                                                          \
        char* kb_ = smem + (BUF_) * BUF;                                                                       \
        char* vb_ = kb_ + KT_BYTES;                                                                            \
        _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                                      \
            *reinterpret_cast<u32x4*>(kb_ + kswz<HD>(krow[i], kkc[i])) = rk[i];                                \
            char* vp_ = vb_ + vrow[i] * VROW + vkc[i] * 16;                                                    \
            *reinterpret_cast<u32x2*>(vp_) = u32x2{rv[i][0], rv[i][1]};                                     \
            *reinterpret_cast<u32x2*>(vp_ + 8) = u32x2{rv[i][2], rv[i][3]};                                 \
        }                                                                                                      \
    }

    f32x16_t oacc[HD / 32];
#pragma unroll
    for (int i = 0; i < HD / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const float sc = a.scale * 1.4426950408889634f;

    ATTN_GLOAD(0)
    ATTN_LSTORE(0)
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) ATTN_GLOAD(t + 1)
        const char* kb = smem + (t & 1) * BUF;
        const char* vb = kb + KT_BYTES;
        const int kv0 = t * BKV;
        if (active) {
            f32x16_t S[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[st][r] = 0.f;
                const int row = st * 32 + c32;
#pragma unroll
                for (int s = 0; s < HD / 16; ++s) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + kswz<HD>(row, s * 2 + h));
                    S[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], S[st], 0, 0, 0);
                }
            }
            // ---- online softmax (base 2; the score scale is folded into the exp2 argument: p = 2^(s*sc - m)) ----
            const bool edge = (kv0 + BKV > a.nkv_rows) || (a.causal && (kv0 + BKV - 1 > q0 + off));
            float mloc = -1e30f;
            if (edge) {
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const bool ok = kv < a.nkv_rows && (!a.causal || kv <= qabs + off);
                        S[st][r] = ok ? S[st][r] : -1e30f;
                    }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, S[st][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float mnew = fmaxf(m, mloc * sc);           // running max in the scaled (base-2) domain
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            float lsum = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(S[st][r], sc, -mnew));   // raw v_exp_f32 (exp2f() adds a denormal-range fix-up: 5 ops)
                    lsum += p;
                    S[st][r] = p;
                }
            l = l * alpha + lsum;
#pragma unroll
            for (int i = 0; i < HD / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            // ---- O^T += V^T . P ----
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int st = ks >> 1, rb = (ks & 1) * 8;
                union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
                for (int i = 0; i < 4; ++i) pf.u[i] = pack2bf(S[st][rb + 2 * i], S[st][rb + 2 * i + 1]);
#pragma unroll
                for (int ht = 0; ht < HD / 32; ++ht) {
                    const char* vp = vb + (ht * 32 + c32) * VROW + (ks * 16 + 4 * h) * 2;
                    union { bf16x8_t v; uint2 u[2]; } vf;
                    vf.u[0] = *reinterpret_cast<const uint2*>(vp);
                    vf.u[1] = *reinterpret_cast<const uint2*>(vp + 16);
                    oacc[ht] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, oacc[ht], 0, 0, 0);
                }
            }
        }
        if (more) ATTN_LSTORE((t + 1) & 1)
        __syncthreads();
    }

    if (active && qabs < a.nq_rows) {
        const float lt = l + __shfl_xor(l, 32, 64);
        const float inv = 1.f / lt;
        bf16_t* op = a.O + (size_t)b * a.o_bs + (size_t)qh * a.o_hs + (size_t)qabs * a.o_rs;
#pragma unroll
        for (int ht = 0; ht < HD / 32; ++ht)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = ht * 32 + 8 * rg + 4 * h;
                uint2 o;
                o.x = pack2bf(oacc[ht][rg * 4 + 0] * inv, oacc[ht][rg * 4 + 1] * inv);
                o.y = pack2bf(oacc[ht][rg * 4 + 2] * inv, oacc[ht][rg * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + d) = o;
            }
    }
}

// V [rows, HD] (row stride src_rs) -> V^T [HD, rows_pad] (row stride dst_rs), zero-filled for rows >= n.
// One block per 64-row tile: 16-byte loads -> LDS -> 16-byte stores along the token axis.
template <int HD>
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ src, long src_bs, long src_hs, int src_rs,
                                                          bf16_t* __restrict__ dst, long dst_bs, long dst_hs, int dst_rs,
                                                          int n) {
    constexpr int LROW = HD + 2;   // elements; odd dword stride -> column reads spread over banks
    __shared__ bf16_t tile[64 * LROW];
    const int t0 = blockIdx.x * 64, hh = blockIdx.y, b = blockIdx.z;
    const bf16_t* s = src + (size_t)b * src_bs + (size_t)hh * src_hs;
    bf16_t* d = dst + (size_t)b * dst_bs + (size_t)hh * dst_hs;
    constexpr int CPR = HD / 8;
    for (int c = threadIdx.x; c < 64 * CPR; c += 256) {
        const int row = c / CPR, kc = c % CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t0 + row < n) v = *reinterpret_cast<const uint4*>(s + (size_t)(t0 + row) * src_rs + kc * 8);
        uint32_t* tp = reinterpret_cast<uint32_t*>(&tile[row * LROW + kc * 8]);
        tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < HD * 8; c += 256) {
        const int dd = c >> 3, tc = c & 7;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t lo = tile[(tc * 8 + 2 * e) * LROW + dd], hi = tile[(tc * 8 + 2 * e + 1) * LROW + dd];
            o[e] = lo | (hi << 16);
        }
        *reinterpret_cast<uint4*>(d + (size_t)dd * dst_rs + t0 + tc * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// HD = 128 needs 238 VGPRs (two waves per SIMD); HD = 64 needs 140, and its loop is softmax-VALU heavy (32 exp2 per lane per
// key tile against 16 MFMAs), so a third wave per SIMD gives the VALU and matrix pipes more independent work to overlap.
template <int HD, bool GQA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD == 64 ? 3 : 2, HD == 64 ? 3 : 2))) void attn_kernel(AttnArgs a) {
    attn_body<HD, GQA>(a);
}

template <int HD, bool GQA>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    constexpr int BUF = BKV * HD * 2 + HD * VROW;
    const size_t lds = 2 * BUF;
    static bool done = false;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<HD, GQA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
    }
    const int nqt = (a.nq_rows + 31) / 32;
    dim3 grid(GQA ? nqt : (nqt + 3) / 4, a.kv_heads, a.batch);
    hipLaunchKernelGGL((attn_kernel<HD, GQA>), grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
}  // namespace

int launch_attn_vit(const AttnArgs& a, hipStream_t s) {
    if (a.heads != a.kv_heads || a.nq_rows <= 0 || a.nkv_rows <= 0 || (a.v_rs % 64)) return TRACE_ERR_ARG;
    return launch_attn<64, false>(a, s);
}

int launch_attn_prefill(const AttnArgs& a, hipStream_t s) {
    if (a.heads != 4 * a.kv_heads || a.nq_rows <= 0 || a.nkv_rows < a.nq_rows || (a.v_rs % 64)) return TRACE_ERR_ARG;
    return launch_attn<128, true>(a, s);
}

int launch_transpose_v(const bf16_t* src, long src_bs, long src_hs, int src_rs, bf16_t* dst, long dst_bs, long dst_hs,
                       int dst_rs, int n, int hd, int heads, int batch, hipStream_t s) {
    if (dst_rs % 64 || dst_rs < n || (hd != 64 && hd != 128)) return TRACE_ERR_ARG;
    dim3 grid((n + 63) / 64, heads, batch);
    if (hd == 64) hipLaunchKernelGGL(transpose_v_kernel<64>, grid, dim3(256), 0, s, src, src_bs, src_hs, src_rs, dst, dst_bs, dst_hs, dst_rs, n);
    else hipLaunchKernelGGL(transpose_v_kernel<128>, grid, dim3(256), 0, s, src, src_bs, src_hs, src_rs, dst, dst_bs, dst_hs, dst_rs, n);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
