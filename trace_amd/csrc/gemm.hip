// bf16 MFMA GEMM for the ViT and LLM-prefill projections:  C[M,N] = A[M,K] . W[N,K]^T (+ epilogue)
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear [out,in] layout), which is
// the natural MFMA layout: every lane's 8-element fragment is one 16-byte read.  Tile 128x128x64, 4 waves
// (2x2, 64x64 each, 16x16x32 MFMA, fp32 accumulate), double-buffered XOR-swizzled LDS (conflict-free
// ds_read_b128), one barrier per K-tile, register-staged prefetch of the next tile under the MFMAs.
// Operands are fed swapped (D^T = W.A^T) so each lane ends up with 4 consecutive output columns of one row;
// the tile is then staged through LDS and leaves as full 16-byte row-contiguous stores.
// Blocks are remapped XCD-aware (common.h) with n fastest so the 8 (or more) N-tiles of one A row-panel
// run on one XCD and hit its L2.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_TILE = BM * BK * 2;          // 16 KiB per operand tile
constexpr int EPI_STRIDE = BN * 2 + 16;        // bytes per staged output row (padded)

__device__ __forceinline__ int swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf0: A | W][buf1: A | W]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = t / ntn, tn = t - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // per-thread staging coordinates: 4 chunks of 16 B per operand
    const bf16_t* ag[4];
    const bf16_t* wg[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256, row = c >> 3, kc = c & 7;
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        ag[i] = p.A + (size_t)am * p.lda + kc * 8;
        wg[i] = p.W + (size_t)(n0 + row) * p.ldw + kc * 8;
        soff[i] = swz(row, kc);
    }
    uint4 ra[4], rw[4];
    const int nk = p.K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(ag[i]);
        rw[i] = *reinterpret_cast<const uint4*>(wg[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(smem + soff[i]) = ra[i];
        *reinterpret_cast<uint4*>(smem + LDS_TILE + soff[i]) = rw[i];
    }
    __syncthreads();

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes within a tile) for k-step 0; k-step 1 flips chunk bit 2
    int aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = swz(wm * 64 + i * 16 + r, g);
        woff[i] = swz(wn * 64 + i * 16 + r, g);
    }

    for (int kt = 0; kt < nk; ++kt) {
        const char* sa = smem + (kt & 1) * 2 * LDS_TILE;
        const char* sw = sa + LDS_TILE;
        const bool more = kt + 1 < nk;
        if (more) {
            const int ko = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const uint4*>(ag[i] + ko);
                rw[i] = *reinterpret_cast<const uint4*>(wg[i] + ko);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[i] ^ (ks << 6)));
                wf[i] = *reinterpret_cast<const bf16x8_t*>(sw + (woff[i] ^ (ks << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* da = smem + ((kt + 1) & 1) * 2 * LDS_TILE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<uint4*>(da + soff[i]) = ra[i];
                *reinterpret_cast<uint4*>(da + LDS_TILE + soff[i]) = rw[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: registers (bias / activation, fp32) -> bf16 -> LDS row-major -> 16-byte stores ----
    // acc[i][j][q] = C[m = m0 + wm*64 + i*16 + r][n = n0 + wn*64 + j*16 + g*4 + q]
    constexpr bool GLU = (EPI == EPI_SWIGLU);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mrow = wm * 64 + i * 16 + r;
        if (!GLU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nl = wn * 64 + j * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = acc[i][j][q];
                    if (p.bias) x += bf2f(p.bias[n0 + nl + q]);
                    if (EPI == EPI_QUICKGELU) x = x / (1.f + __expf(-1.702f * x));
                    v[q] = x;
                }
                uint2 pk = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                *reinterpret_cast<uint2*>(smem + mrow * EPI_STRIDE + nl * 2) = pk;
            }
        } else {
            // W rows interleaved per 16: tile j even = gate rows, j odd = up rows of the same 16 outputs
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int nl = wn * 32 + jj * 16 + g * 4;          // output column within the 64-wide half tile
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gt = acc[i][2 * jj][q], up = acc[i][2 * jj + 1][q];
                    v[q] = gt / (1.f + __expf(-gt)) * up;
                }
                uint2 pk = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                *reinterpret_cast<uint2*>(smem + mrow * EPI_STRIDE + nl * 2) = pk;
            }
        }
    }
    __syncthreads();
    constexpr int OUTW = GLU ? BN / 2 : BN;                 // output columns of this tile
    constexpr int CPR = OUTW / 8;                           // 16-byte chunks per row
    const int on0 = GLU ? n0 / 2 : n0;
    for (int c = tid; c < BM * CPR; c += 256) {
        const int row = c / CPR, ch = c - row * CPR;
        const int m = m0 + row;
        if (m >= p.M) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * EPI_STRIDE + ch * 16);
        if (EPI == EPI_RESIDUAL) {
            const uint4 rr = *reinterpret_cast<const uint4*>(p.R + (size_t)m * p.ldr + on0 + ch * 8);
            v.x = pack2bf(bflo(v.x) + bflo(rr.x), bfhi(v.x) + bfhi(rr.x));
            v.y = pack2bf(bflo(v.y) + bflo(rr.y), bfhi(v.y) + bfhi(rr.y));
            v.z = pack2bf(bflo(v.z) + bflo(rr.z), bfhi(v.z) + bfhi(rr.z));
            v.w = pack2bf(bflo(v.w) + bflo(rr.w), bfhi(v.w) + bfhi(rr.w));
        }
        *reinterpret_cast<uint4*>(p.C + (size_t)m * p.ldc + on0 + ch * 8) = v;
    }
}


// =========================================================================================================
// v2: direct-to-LDS (global_load_lds, 16 B/lane) double-buffered tiles, BMxBNx64, (WM x WN) waves.
// The LDS image is lane-linear per wave instruction (HW: M0 base + lane*16), so the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane *source* address (same involution on
// the read side).  One barrier per K-tile: tile kt+1 streams into the other buffer while tile kt feeds the MFMAs.
// Tile order: XCD-contiguous, then grouped (8 row panels x all column panels) so the ~32 workgroups resident on
// one XCD share A row-panels and W column-panels through that XCD's L2.
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_kernel(GemmArgs p) {
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int A_IT = BM * 8 / NTHR, W_IT = BN * 8 / NTHR;
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    constexpr int OSTRIDE = BN * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wid / WN, wn = wid % WN;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    int t = xcd_remap(blockIdx.x, ntm * ntn);
    int tm, tn;
    {
        constexpr int GM = 8;
        const int per_group = GM * ntn;
        const int gid = t / per_group, first = gid * GM;
        const int gsz = min(ntm - first, GM);
        const int in_g = t - gid * per_group;
        tm = first + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16_t* asrc[A_IT];
    const bf16_t* wsrc[W_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int slot = i * NTHR + tid, row = slot >> 3, kc = (slot & 7) ^ ((row >> 1) & 7);
        const int am = min(m0 + row, p.M - 1);
        asrc[i] = p.A + (size_t)am * p.lda + kc * 8;
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int slot = i * NTHR + tid, row = slot >> 3, kc = (slot & 7) ^ ((row >> 1) & 7);
        wsrc[i] = p.W + (size_t)(n0 + row) * p.ldw + kc * 8;
    }
    auto issue = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE;
        const int ko = kt * BK;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sa + (i * NTHR + wid * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sa + A_BYTES + (i * NTHR + wid * 64) * 16), 16, 0, 0);
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int aoff[TM], woff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = swz(wm * (BM / WM) + i * 16 + r, g);
#pragma unroll
    for (int j = 0; j < TN; ++j) woff[j] = swz(wn * (BN / WN) + j * 16 + r, g);

    const int nk = p.K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // tile kt landed (vmcnt(0) + barrier); everyone is done with tile kt-1
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[i] ^ (ks << 6)));
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sw + (woff[j] ^ (ks << 6)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    // ---- epilogue (same scheme as v1): fp32 bias/activation -> bf16 -> LDS rows -> 16-byte stores ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = wm * (BM / WM) + i * 16 + r;
        if (!GLU) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * (BN / WN) + j * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = acc[i][j][q];
                    if (p.bias) x += bf2f(p.bias[n0 + nl + q]);
                    if (EPI == EPI_QUICKGELU) x = x / (1.f + __expf(-1.702f * x));
                    v[q] = x;
                }
                *reinterpret_cast<uint2*>(smem + mrow * OSTRIDE + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < TN / 2; ++jj) {
                const int nl = wn * (BN / WN / 2) + jj * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gt = acc[i][2 * jj][q], up = acc[i][2 * jj + 1][q];
                    v[q] = gt / (1.f + __expf(-gt)) * up;
                }
                *reinterpret_cast<uint2*>(smem + mrow * OSTRIDE + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
    }
    __syncthreads();
    constexpr int OUTW = GLU ? BN / 2 : BN;
    constexpr int CPR = OUTW / 8;
    const int on0 = GLU ? n0 / 2 : n0;
    for (int c = tid; c < BM * CPR; c += NTHR) {
        const int row = c / CPR, ch = c - row * CPR;
        const int m = m0 + row;
        if (m >= p.M) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * OSTRIDE + ch * 16);
        if (EPI == EPI_RESIDUAL) {
            const uint4 rr = *reinterpret_cast<const uint4*>(p.R + (size_t)m * p.ldr + on0 + ch * 8);
            v.x = pack2bf(bflo(v.x) + bflo(rr.x), bfhi(v.x) + bfhi(rr.x));
            v.y = pack2bf(bflo(v.y) + bflo(rr.y), bfhi(v.y) + bfhi(rr.y));
            v.z = pack2bf(bflo(v.z) + bflo(rr.z), bfhi(v.z) + bfhi(rr.z));
            v.w = pack2bf(bflo(v.w) + bflo(rr.w), bfhi(v.w) + bfhi(rr.w));
        }
        *reinterpret_cast<uint4*>(p.C + (size_t)m * p.ldc + on0 + ch * 8) = v;
    }
}

template <int BM, int BN, int WM, int WN>
int launch_glds(const GemmArgs& p, int epi, hipStream_t s) {
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int OBYTES = BM * (BN * 2 + 16);
    constexpr size_t lds = (2 * STAGE > OBYTES) ? 2 * STAGE : OBYTES;
    constexpr int NTHR = WM * WN * 64;
    const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_QUICKGELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WM, WN, EPI_SWIGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_NONE>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        case EPI_RESIDUAL: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_RESIDUAL>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        case EPI_QUICKGELU: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_QUICKGELU>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, EPI_SWIGLU>), dim3(nblk), dim3(NTHR), lds, s, p); break;
        default: return TRACE_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

}  // namespace

int g_gemm_variant = 0;   // 0 = auto, 1 = v1 128^2 register-staged, 2 = glds 128^2, 3 = glds 256^2 (tests / microbench)

int launch_gemm_bf16(const GemmArgs& p, int epi, hipStream_t s) {
    if (p.M <= 0 || p.N % BN || p.K % BK || p.K < BK) return TRACE_ERR_ARG;
    if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 8)) return TRACE_ERR_ARG;
    if (epi == EPI_RESIDUAL && (!p.R || (p.ldr % 8))) return TRACE_ERR_ARG;
    {
        int v = g_gemm_variant;
        if (v == 0) {
            const long blocks256 = (long)((p.M + 255) / 256) * (p.N / 256);
            v = (p.N % 256 == 0 && p.M >= 1024 && blocks256 >= 200) ? 3 : 2;
        }
        if (v == 3 && p.N % 256 == 0) return launch_glds<256, 256, 2, 4>(p, epi, s);
        if (v == 2) return launch_glds<128, 128, 2, 2>(p, epi, s);
    }
    const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
    const size_t lds = 4 * LDS_TILE;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_QUICKGELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_SWIGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_NONE>, dim3(nblk), dim3(256), lds, s, p); break;
        case EPI_RESIDUAL: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_RESIDUAL>, dim3(nblk), dim3(256), lds, s, p); break;
        case EPI_QUICKGELU: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_QUICKGELU>, dim3(nblk), dim3(256), lds, s, p); break;
        case EPI_SWIGLU: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_SWIGLU>, dim3(nblk), dim3(256), lds, s, p); break;
        default: return TRACE_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
