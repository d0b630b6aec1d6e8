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

}  // namespace

int launch_gemm_bf16(const GemmArgs& p, int epi, hipStream_t s) {
    if (p.M <= 0 || p.N % BN || p.K % BK || p.K < BK) return TRACE_ERR_ARG;
    if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 8)) return TRACE_ERR_ARG;
    if (epi == EPI_RESIDUAL && (!p.R || (p.ldr % 8))) return TRACE_ERR_ARG;
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
