// Persistent variant of the 256x256 bf16 MFMA GEMM (gemm.hip) that hides a tile's epilogue under the NEXT tile's K loop.
//
// gemm.hip's per-tile anatomy on the K = 1024 ViT shapes is ~27 us of K loop (at the 1.26 PFLOP/s the matrix pipe +
// LDS-DMA path sustain) + ~8 us outside it: epilogue math (QuickGELU = two quarter-rate transcendentals x 128 values per
// lane), LDS staging, stores, and the first K-tile's load latency.  Here one workgroup per CU walks tiles
// blockIdx, blockIdx + grid, ...; at the end of a tile it only adds the bias and packs the accumulators to bf16
// ("parked": 128 -> 64 registers; exactly what gemm.hip's epilogue would have stored for EPI_NONE / EPI_RESIDUAL, one
// extra rounding before the activation for EPI_QUICKGELU), and the first 16 K-tiles of the next tile carry the parked
// tile out, one 32-row strip per two K-tiles: K-tile 2s activates strip s and writes it to a 17 KB LDS staging strip,
// K-tile 2s+1 (behind the barrier the K loop has anyway) reads full rows back and issues the 16-byte stores.  The
// first K-tile of the next tile is requested during the last K-tile of the current one, so the prologue disappears too.
// NOT BUILT INTO THE LIBRARY — see experiments/README.md for what was measured (slower than gemm.hip: register spills;
// EPI_QUICKGELU wrong after a workgroup's first tile).  Needs K >= 512 (8 K-tiles to carry the 4 parked strips).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4, NTHR = 512;
constexpr int TM = BM / WM / 16, TN = BN / WN / 16;          // 8 x 4 MFMA tiles per wave
constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
constexpr int A_IT = BM * 8 / NTHR, W_IT = BN * 8 / NTHR;    // 4 + 4 LDS-DMA pieces per thread per K-tile
constexpr int SROW = BN * 2 + 16;                            // staging row pitch (bytes)
constexpr int STG_OFF = 2 * STAGE, LDS_BYTES = STG_OFF + 32 * SROW;
constexpr int NSTRIP = TM;                                   // strip s = MFMA row-tile s of both wave rows = 32 full rows
constexpr int NPARK = 4;                                     // strips NSTRIP-NPARK.. are parked in registers; the first ones leave at tile end

__device__ __forceinline__ int swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(NTHR) void gemm_pipe_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wid / WN, wn = wid % WN;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM, ntiles = ntm * ntn;
    const int nk = p.K / BK;
    char* stg = smem + STG_OFF;

    auto tile_origin = [&](int vb, int& m0, int& n0) {
        constexpr int GM = 8;
        const int t = xcd_remap(vb, ntiles);
        const int per_group = GM * ntn;
        const int gid = t / per_group, first = gid * GM;
        const int gsz = min(ntm - first, GM);
        const int in_g = t - gid * per_group;
        m0 = (first + in_g % gsz) * BM;
        n0 = (in_g / gsz) * BN;
    };
    // LDS-DMA sources of the tile being fetched (a tile's last K-tile already fetches for the next tile).  Piece i of A is row
    // i*64 + tid/8 of the tile, 16-byte chunk kc (the XOR swizzle lives in the SOURCE chunk; (row >> 1) & 7 does not depend on i);
    // kept as one row index + one W element offset instead of eight pointers — the register file is full (see below)
    const int trow = tid >> 3, kc8 = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    int src_m0 = 0, src_w0 = 0;        // first A row of the tile being fetched; element offset of its W row `trow`
    auto set_sources = [&](int m0, int n0) { src_m0 = m0 + trow; src_w0 = (n0 + trow) * p.ldw + kc8; };
    auto issue_piece = [&](int stage, int ko, int piece) {
        char* base = smem + stage * STAGE;
        if (piece < A_IT) {
            const int row = min(src_m0 + piece * 64, p.M - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + (size_t)(unsigned)(row * p.lda + kc8) + ko),
                                             (__attribute__((address_space(3))) void*)(base + (piece * NTHR + wid * 64) * 16), 16, 0, 0);
        } else {
            const int i = piece - A_IT;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.W + (size_t)(unsigned)(src_w0 + i * 64 * p.ldw) + ko),
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + (i * NTHR + wid * 64) * 16), 16, 0, 0);
        }
    };

    f32x4_t acc[TM][TN];
    uint2 parked[NPARK][TN];           // last NPARK strips of the previous tile: bf16(acc + bias); parked[0] is the next strip to leave
    int pm0 = 0, pn0 = 0;              // its origin
    bool have_parked = false;
    uint4 rres[2];                     // residual pieces of the strip in flight (EPI_RESIDUAL)
    // fragment addresses: tile i of a wave is 16 rows = 2048 bytes further down and ((row >> 1) & 7) is the same for all of
    // them, so one swizzled base per operand and k-half + immediates (ds_read offsets) replace twelve address registers
    const int abase = swz(wm * (BM / WM) + r, g), wbase = swz(wn * (BN / WN) + r, g);

    // ---- strip phases of the parked tile -----------------------------------------------------------------
    // phase A: activate parked[0] and write this lane's 4 x 8 bytes of the strip's 32 x 256 image; request the residual pieces
    auto strip_write = [&](const uint2 (&vals)[TN], int s) {
        const int srow = wm * 16 + r;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            uint2 v = vals[j];
            if (EPI == EPI_QUICKGELU) {
                float x[4] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y)};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    x[q] = x[q] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x[q]));   // x*sigmoid(1.702x)
                v = make_uint2(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]));
            }
            *reinterpret_cast<uint2*>(stg + srow * SROW + (wn * (BN / WN) + j * 16 + g * 4) * 2) = v;
        }
        if (EPI == EPI_RESIDUAL) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = it * NTHR + tid, sr = c >> 5, ch = c & 31;
                const int m = min(pm0 + (sr >> 4) * (BM / WM) + s * 16 + (sr & 15), p.M - 1);
                rres[it] = *reinterpret_cast<const uint4*>(p.R + (size_t)m * p.ldr + pn0 + ch * 8);
            }
        }
    };
    // phase B: full rows back out of LDS, 16-byte non-temporal stores
    auto strip_store = [&](int s) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = it * NTHR + tid, sr = c >> 5, ch = c & 31;
            const int m = pm0 + (sr >> 4) * (BM / WM) + s * 16 + (sr & 15);
            uint4 v = *reinterpret_cast<const uint4*>(stg + sr * SROW + ch * 16);
            if (EPI == EPI_RESIDUAL) {
                const uint4 rr = rres[it];
                v.x = pack2bf(bflo(v.x) + bflo(rr.x), bfhi(v.x) + bfhi(rr.x));
                v.y = pack2bf(bflo(v.y) + bflo(rr.y), bfhi(v.y) + bfhi(rr.y));
                v.z = pack2bf(bflo(v.z) + bflo(rr.z), bfhi(v.z) + bfhi(rr.z));
                v.w = pack2bf(bflo(v.w) + bflo(rr.w), bfhi(v.w) + bfhi(rr.w));
            }
            if (m < p.M) {
                typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
                __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + pn0 + ch * 8));
            }
        }
    };
    auto rotate_parked = [&]() {
#pragma unroll
        for (int i = 0; i + 1 < NPARK; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) parked[i][j] = parked[i + 1][j];
    };

    // ---- one K-tile: barrier, 64 MFMAs per wave with the next K-tile's 8 LDS-DMA pieces spread between the MFMA groups,
    //      plus (PH 1 / 2) one strip phase of the parked tile ------------------------------------------------------
    int gk = 0;                        // K-tiles done by this workgroup (stage parity runs across tiles)
    auto ktile = [&](auto ph, int s, int pf_ko) {
        constexpr int PH = decltype(ph)::value;
        __syncthreads();               // K-tile gk landed; everyone is done with K-tile gk-1 and with the staging strip's last phase
        const char* sa = smem + (gk & 1) * STAGE;
        const char* sw = sa + A_BYTES;
        const int pf_stage = (gk + 1) & 1;
        if (PH == 1) strip_write(parked[0], s);
        if (PH == 2) { strip_store(s); rotate_parked(); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* fa = sa + (abase ^ (ks << 6));
            const char* fw = sw + (wbase ^ (ks << 6));
            bf16x8_t wf[TN], ac[2];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(fw + j * 2048);
            __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
#pragma unroll
            for (int ip = 0; ip < TM / 2; ++ip) {
                ac[0] = *reinterpret_cast<const bf16x8_t*>(fa + (2 * ip) * 2048);
                ac[1] = *reinterpret_cast<const bf16x8_t*>(fa + (2 * ip + 1) * 2048);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[2 * ip + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], ac[ii], acc[2 * ip + ii][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                issue_piece(pf_stage, pf_ko, ks * (TM / 2) + ip);       // unconditional: keeps the K-tile one basic block
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        ++gk;
    };
    typedef std::integral_constant<int, 0> PH0;
    typedef std::integral_constant<int, 1> PH1;
    typedef std::integral_constant<int, 2> PH2;

    int vb = blockIdx.x;
    int m0 = 0, n0 = 0;
    if (vb < ntiles) {
        tile_origin(vb, m0, n0);
        set_sources(m0, n0);
#pragma unroll
        for (int q = 0; q < A_IT + W_IT; ++q) issue_piece(0, 0, q);
    }
    for (; vb < ntiles; vb += gridDim.x) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int cm0 = m0, cn0 = n0;
        // what K-tile kt prefetches: the next K-tile of this tile, or K-tile 0 of the next tile (sources switched first).  The
        // workgroup's very last K-tile re-requests K-tile 0 of its own tile (64 KB of L2 reads nobody uses) so that no K-tile
        // needs a branch around its loads.
        auto step = [&](auto ph, int s, int kt) {
            int ko = (kt + 1) * BK;
            if (kt + 1 == nk) {
                ko = 0;
                if (vb + (int)gridDim.x < ntiles) { tile_origin(vb + gridDim.x, m0, n0); set_sources(m0, n0); }
            }
            ktile(ph, s, ko);
        };
        int kt = 0;
        if (have_parked) {
            for (int s = NSTRIP - NPARK; s < NSTRIP; ++s) { step(PH1{}, s, kt++); step(PH2{}, s, kt++); }
        }
        for (; kt < nk; ++kt) step(PH0{}, 0, kt);
        // tile end: bias in fp32, round to bf16 (gemm.hip's epilogue rounding for EPI_NONE / EPI_RESIDUAL); the first strips leave
        // now (their barriers also cover the latency of the next tile's first K-tile, already requested), the rest are parked
        pm0 = cm0; pn0 = cn0;
        float bv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            uint2 b2 = make_uint2(0u, 0u);
            if (p.bias) b2 = *reinterpret_cast<const uint2*>(p.bias + cn0 + wn * (BN / WN) + j * 16 + g * 4);
            bv[j][0] = bflo(b2.x); bv[j][1] = bfhi(b2.x); bv[j][2] = bflo(b2.y); bv[j][3] = bfhi(b2.y);
        }
#pragma unroll
        for (int i = 0; i < NSTRIP; ++i) {
            uint2 v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                v[j] = make_uint2(pack2bf(acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1]), pack2bf(acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]));
            if (i < NSTRIP - NPARK) {
                __syncthreads();
                strip_write(v, i);
                __syncthreads();
                strip_store(i);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) parked[i - (NSTRIP - NPARK)][j] = v[j];
            }
        }
        have_parked = true;
    }
    // ---- the last tile's parked strips leave synchronously ----
    if (have_parked) {
        for (int s = NSTRIP - NPARK; s < NSTRIP; ++s) {
            __syncthreads();
            strip_write(parked[0], s);
            __syncthreads();
            strip_store(s);
            rotate_parked();
        }
    }
}

}  // namespace

// returns TRACE_ERR_ARG when the shape is not one this variant handles (the caller falls back to gemm.hip's kernels)
int launch_gemm_pipe(const GemmArgs& p, int epi, hipStream_t s) {
    if (p.M < 1 || p.N % BN || p.K % BK || p.K / BK < 2 * NPARK) return TRACE_ERR_ARG;
    if (epi != EPI_NONE && epi != EPI_RESIDUAL && epi != EPI_QUICKGELU) return TRACE_ERR_ARG;
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return TRACE_ERR_ARG;      // 32-bit element offsets
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;                                          // whole XCD rounds: blockIdx + k*grid stays on one XCD's tile range
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pipe_kernel<EPI_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pipe_kernel<EPI_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pipe_kernel<EPI_QUICKGELU>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    const int grid = ntiles < ncu ? ntiles : ncu;
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL(gemm_pipe_kernel<EPI_NONE>, dim3(grid), dim3(NTHR), LDS_BYTES, s, p); break;
        case EPI_RESIDUAL: hipLaunchKernelGGL(gemm_pipe_kernel<EPI_RESIDUAL>, dim3(grid), dim3(NTHR), LDS_BYTES, s, p); break;
        default: hipLaunchKernelGGL(gemm_pipe_kernel<EPI_QUICKGELU>, dim3(grid), dim3(NTHR), LDS_BYTES, s, p); break;
    }
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
