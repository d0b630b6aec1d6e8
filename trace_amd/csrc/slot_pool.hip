// SpatialSlotPool (reference trace/model/multimodal_projector/builder.py:411-467), everything before `readout`:
//   x = LayerNorm_eps1e-6(feats)           (timm LayerNorm, :451)
//   x = x*cos + rotate_half(x)*sin         (channel-axis RoPE, position = patch index, :289-359,:453-455)
//   logits[n,s] = x[n,:] . slots[:,s]      (:457)      P = softmax over the n patches (:458)
//   res[s,:] = sum_n P[n,s] * x[n,:]       (:462)
// Round 2: the first version (one 4-wave workgroup per frame, all arithmetic on fp32 VALU: ~1200 instructions per patch row) used
// half the CUs and was VALU-bound — 454 us per 128 frames = 0.35 TB/s, 4 % of the HBM roofline its 151 MB of input would allow.
// Now each frame is split over NPART workgroups (1024 for 128 frames), each with a LOCAL softmax (max, sum, weighted sums against its own
// max) merged by a second small kernel (exact: exp(m_part - M) rescaling), and the per-row work is ~420 instructions: row sums by
// v_dot2c_f32_bf16 on the packed input, logits by v_dot2c on the bf16-rounded rotated row against slot pairs held in registers (the
// reference module runs in half precision here too), wave reductions on DPP instead of ds_bpermute shuffles, and
// packed fp32 FMAs for the weighted sums.  Every reduction keeps a fixed order (bit-reproducible).  Output bf16 [T*S, D] for the readout GEMM.
#include "common.h"
#include "kernels.h"

namespace {
constexpr int NS = 8;   // slots

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    v[0] = bflo(u.x); v[1] = bfhi(u.x); v[2] = bflo(u.y); v[3] = bfhi(u.y);
    v[4] = bflo(u.z); v[5] = bfhi(u.z); v[6] = bflo(u.w); v[7] = bfhi(u.w);
}

typedef __attribute__((ext_vector_type(2))) float f2_t;
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return dot2_el(a, b, c);
}
constexpr int NPART = 8;      // workgroups per frame

// part g of frame t: patch rows [g*RP, (g+1)*RP).  Partial outputs: pm / pl [T][NPART][8] (local max, local sum of exp),
// pres [T][NPART][8][D] fp32 (sum_p exp(l - m_local) * x_hat[p, :]).
__global__ __launch_bounds__(256) void slot_pool_part_kernel(const bf16_t* __restrict__ feats, long frame_stride, int row_stride,
                                                             const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b,
                                                             const bf16_t* __restrict__ slots, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, float* __restrict__ pm, float* __restrict__ pl,
                                                             float* __restrict__ pres, int n, int D, int RP, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H2 = D >> 1, NL = D >> 4;          // lane l < NL owns d in [8l, 8l+8) and H2 + [8l, 8l+8): the RoPE pairs (d, d + H2)
    float* s_logit = reinterpret_cast<float*>(smem);                  // [RP][8]
    float* s_stat = s_logit + (size_t)RP * NS;                        // [RP][2] mean, rstd
    float* s_red = s_stat + (size_t)((RP + 1) & ~1) * 2;              // [8] local max (RP rounded up to even: s_res below is read as float4)
    float* s_res = s_red + 16;                                        // [8][D]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.x, t = blockIdx.y;
    const int p_beg = g * RP, p_end = min(n, p_beg + RP), np = max(0, p_end - p_beg);
    const bf16_t* fr = feats + (size_t)t * frame_stride;
    const bool on = lane < NL;
    const int lo = on ? lane : 0;

    float w1[8], w2[8], b1[8], b2[8];
    {
        uint4 u;
        u = *reinterpret_cast<const uint4*>(ln_w + lo * 8); unpack8(u, w1);
        u = *reinterpret_cast<const uint4*>(ln_w + H2 + lo * 8); unpack8(u, w2);
        u = *reinterpret_cast<const uint4*>(ln_b + lo * 8); unpack8(u, b1);
        u = *reinterpret_cast<const uint4*>(ln_b + H2 + lo * 8); unpack8(u, b2);
    }
    // one patch row's inputs: the lane's 16 features and its 8 cos / 8 sin values.  The loops below keep the NEXT row's loads in flight
    // while the current row is computed (with 2 workgroups per CU nothing else hides the ~1.5 us the loads take)
    struct RowIn { uint4 u1, u2; float4 c0, c1, s0, s1; };
    auto load_row = [&](int p) {
        RowIn q;
        const bf16_t* xr = fr + (size_t)p * row_stride;
        q.u1 = *reinterpret_cast<const uint4*>(xr + lo * 8);
        q.u2 = *reinterpret_cast<const uint4*>(xr + H2 + lo * 8);
        const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)p * H2 + lo * 8);
        const float4* sp_ = reinterpret_cast<const float4*>(sin_t + (size_t)p * H2 + lo * 8);
        q.c0 = cp[0]; q.c1 = cp[1]; q.s0 = sp_[0]; q.s1 = sp_[1];
        return q;
    };
    auto rotated = [&](const RowIn& q, float mean, float rstd, float (&r1)[8], float (&r2)[8]) {
        float x1[8], x2[8];
        unpack8(q.u1, x1); unpack8(q.u2, x2);
        const float cs[8] = {q.c0.x, q.c0.y, q.c0.z, q.c0.w, q.c1.x, q.c1.y, q.c1.z, q.c1.w};
        const float sn[8] = {q.s0.x, q.s0.y, q.s0.z, q.s0.w, q.s1.x, q.s1.y, q.s1.z, q.s1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = (x1[e] - mean) * rstd * w1[e] + b1[e];
            const float b = (x2[e] - mean) * rstd * w2[e] + b2[e];
            r1[e] = a * cs[e] - b * sn[e];
            r2[e] = b * cs[e] + a * sn[e];
        }
    };

    // ---------------- pass 1: LN statistics + logits ----------------
    {
        // slot pairs for v_dot2c: sp[j][k] = (slots[d_j, k], slots[d_j + 1, k]) for the lane's 8 element pairs (4 per half)
        uint32_t sp[8][NS];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = (j < 4 ? 0 : H2) + lo * 8 + (j & 3) * 2;
            const uint4 ra = *reinterpret_cast<const uint4*>(slots + (size_t)d * NS), rb = *reinterpret_cast<const uint4*>(slots + (size_t)(d + 1) * NS);
            const uint32_t A[4] = {ra.x, ra.y, ra.z, ra.w}, Bv[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sp[j][2 * q] = (A[q] & 0xffffu) | (Bv[q] << 16);
                sp[j][2 * q + 1] = (A[q] >> 16) | (Bv[q] & 0xffff0000u);
            }
        }
        RowIn nxt = load_row(min(p_beg + min(wid, max(np - 1, 0)), n - 1));   // (an empty part — p_beg >= n — preloads a valid row it never uses)
        for (int i = wid; i < np; i += 4) {
            const RowIn cur = nxt;
            if (i + 4 < np) nxt = load_row(p_beg + i + 4);
            const uint4 u1 = on ? cur.u1 : make_uint4(0, 0, 0, 0), u2 = on ? cur.u2 : make_uint4(0, 0, 0, 0);
            const uint32_t xp[8] = {u1.x, u1.y, u1.z, u1.w, u2.x, u2.y, u2.z, u2.w};
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { sm = dot2(xp[j], TRACE_EL_ONE2, sm); sq = dot2(xp[j], xp[j], sq); }
            sm = wave_sum_dpp(sm); sq = wave_sum_dpp(sq);
            const float mean = sm / (float)D;
            const float rstd = rsqrtf(fmaxf(sq / (float)D - mean * mean, 0.f) + eps);
            float r1[8], r2[8];
            rotated(cur, mean, rstd, r1, r2);
            uint32_t rp[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { rp[j] = pack2bf(r1[2 * j], r1[2 * j + 1]); rp[4 + j] = pack2bf(r2[2 * j], r2[2 * j + 1]); }
            float lg[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) a = dot2(rp[j], sp[j][k], a);
                lg[k] = on ? a : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) lg[k] = wave_sum_dpp(lg[k]);       // eight independent DPP chains
            if (lane == 0) {
                *reinterpret_cast<float4*>(s_logit + i * NS) = make_float4(lg[0], lg[1], lg[2], lg[3]);
                *reinterpret_cast<float4*>(s_logit + i * NS + 4) = make_float4(lg[4], lg[5], lg[6], lg[7]);
            }
            if (lane == 0) { s_stat[i * 2] = mean; s_stat[i * 2 + 1] = rstd; }
        }
    }
    for (int i = tid; i < NS * D; i += 256) s_res[i] = 0.f;
    __syncthreads();

    // ---------------- local softmax numerators: p = exp(l - m_local) (2 slots per wave) ----------------
    for (int k = wid * 2; k < wid * 2 + 2; ++k) {
        float mx = -1e30f;
        for (int i = lane; i < np; i += 64) mx = fmaxf(mx, s_logit[i * NS + k]);
        mx = wave_max(mx);
        float sme = 0.f;
        for (int i = lane; i < np; i += 64) {
            const float e = __expf(s_logit[i * NS + k] - mx);
            s_logit[i * NS + k] = e;
            sme += e;
        }
        sme = wave_sum(sme);
        if (lane == 0) {
            pm[((size_t)t * NPART + g) * NS + k] = mx;
            pl[((size_t)t * NPART + g) * NS + k] = sme;
        }
    }
    __syncthreads();

    // ---------------- pass 2: sum_p p[p, s] * x_hat[p, :] ----------------
    f2_t a1[NS][4], a2[NS][4];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a1[k][e] = f2_t{0.f, 0.f}; a2[k][e] = f2_t{0.f, 0.f}; }
    if (on) {
        RowIn nxt = load_row(min(p_beg + min(wid, max(np - 1, 0)), n - 1));   // (an empty part — p_beg >= n — preloads a valid row it never uses)
        for (int i = wid; i < np; i += 4) {
            const RowIn cur = nxt;
            if (i + 4 < np) nxt = load_row(p_beg + i + 4);
            float r1[8], r2[8];
            rotated(cur, s_stat[i * 2], s_stat[i * 2 + 1], r1, r2);
            const float4 pa = *reinterpret_cast<const float4*>(s_logit + i * NS), pb = *reinterpret_cast<const float4*>(s_logit + i * NS + 4);
            const float pr[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const f2_t pk = {pr[k], pr[k]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1[k][e] = __builtin_elementwise_fma(pk, f2_t{r1[2 * e], r1[2 * e + 1]}, a1[k][e]);
                    a2[k][e] = __builtin_elementwise_fma(pk, f2_t{r2[2 * e], r2[2 * e + 1]}, a2[k][e]);
                }
            }
        }
    }
    // the four waves add their partial sums one after the other (fixed order: bit-reproducible)
    for (int w = 0; w < 4; ++w) {
        if (on && wid == w) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f2_t* q1 = reinterpret_cast<f2_t*>(s_res + k * D + lane * 8 + 2 * e);
                    f2_t* q2 = reinterpret_cast<f2_t*>(s_res + k * D + H2 + lane * 8 + 2 * e);
                    *q1 += a1[k][e];
                    *q2 += a2[k][e];
                }
        }
        __syncthreads();
    }
    float* out = pres + ((size_t)t * NPART + g) * NS * D;
    for (int i = tid; i < NS * D / 4; i += 256) reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(s_res)[i];
}

// res[t, s, :] = sum_g exp(m_g - M) pres_g / sum_g exp(m_g - M) l_g,  M = max_g m_g  (parts in index order)
__global__ __launch_bounds__(256) void slot_pool_merge_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                              const float* __restrict__ pres, bf16_t* __restrict__ res, int D) {
    const int t = blockIdx.x, tid = threadIdx.x;
    __shared__ float s_w[NPART][NS];
    if (tid < NS) {
        float M = -1e30f;
        for (int g = 0; g < NPART; ++g) M = fmaxf(M, pm[((size_t)t * NPART + g) * NS + tid]);
        float den = 0.f, w[NPART];
        for (int g = 0; g < NPART; ++g) {
            w[g] = __expf(pm[((size_t)t * NPART + g) * NS + tid] - M);
            den += w[g] * pl[((size_t)t * NPART + g) * NS + tid];
        }
        for (int g = 0; g < NPART; ++g) s_w[g][tid] = w[g] / den;
    }
    __syncthreads();
    bf16_t* out = res + (size_t)t * NS * D;
    for (int i = tid; i < NS * D / 2; i += 256) {
        const int k = (2 * i) / D;
        float x0 = 0.f, x1 = 0.f;
        for (int g = 0; g < NPART; ++g) {
            const float2 v = *reinterpret_cast<const float2*>(pres + ((size_t)t * NPART + g) * NS * D + 2 * i);
            x0 += s_w[g][k] * v.x; x1 += s_w[g][k] * v.y;
        }
        reinterpret_cast<uint32_t*>(out)[i] = pack2bf(x0, x1);
    }
}
}  // namespace

// scratch: [T*NPART*8] pm, [T*NPART*8] pl, [T*NPART*8*D] pres (floats): launch_slot_pool_ws_floats(T, D)
size_t launch_slot_pool_ws_floats(int T, int D) { return (size_t)T * NPART * NS * (2 + (size_t)D); }

int launch_slot_pool(const bf16_t* feats, long frame_stride, int row_stride, const bf16_t* ln_w, const bf16_t* ln_b,
                     const bf16_t* slots, const float* cos_t, const float* sin_t, bf16_t* res, int T, int n, int D, int S,
                     float eps, float* ws, size_t ws_floats, hipStream_t s) {
    if (S != NS || D % 16 || D > 1024 || T <= 0 || n <= 0 || (row_stride % 8)) return TRACE_ERR_ARG;
    if (!ws || ws_floats < launch_slot_pool_ws_floats(T, D)) return TRACE_ERR_ARG;
    const int RP = (n + NPART - 1) / NPART;
    const size_t lds = (size_t)RP * NS * 4 + (size_t)((RP + 1) & ~1) * 8 + 64 + (size_t)NS * D * 4;
    if (lds > 160 * 1024) return TRACE_ERR_ARG;
    static LdsGrantSized grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(slot_pool_part_kernel), lds)) return TRACE_ERR_HIP;
    float* pm = ws;
    float* pl = pm + (size_t)T * NPART * NS;
    float* pres = pl + (size_t)T * NPART * NS;
    hipLaunchKernelGGL(slot_pool_part_kernel, dim3(NPART, T), dim3(256), lds, s, feats, frame_stride, row_stride, ln_w, ln_b, slots, cos_t,
                       sin_t, pm, pl, pres, n, D, RP, eps);
    hipLaunchKernelGGL(slot_pool_merge_kernel, dim3(T), dim3(256), 0, s, pm, pl, pres, res, D);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
