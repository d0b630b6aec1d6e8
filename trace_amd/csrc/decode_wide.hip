// The wide decode step's projection chain as ONE persistent launch per layer (round 6).
//
// A wide step (32 .. 128 sequences, engine.hip decode_step_wide) was eight launches per layer: qkv GEMM -> qkv_finish -> attention -> o GEMM -> add_rmsnorm ->
// gate|up GEMM -> down GEMM -> add_rmsnorm.  Everything between two attentions streams weights against 128 activation rows and hands a few MB to the next
// kernel: seven kernels of 6 - 50 us whose fill and drain (the next kernel's first weight tile is requested only after the previous kernel's last workgroup
// has retired) are a fifth of their time.  Here that chain is one launch of one workgroup per CU:
//     o GEMM (split-K partial rows) | add + RMSNorm | gate|up GEMM + SwiGLU | down GEMM (partial rows) | add + RMSNorm | next layer's qkv GEMM | qkv finish
// with a grid barrier (gridbar.h) between two phases.  The GEMM phases are gemm_glds_kernel<128, 128, 2, 2, EPI, false, WT = true, NSTAGE = 4> (gemm.hip)
// statement for statement — same tile walk, same K order, same epilogues — except that a workgroup requests the first three WEIGHT K-tiles of its next
// tile into the LDS ring BEFORE it waits at the barrier (weights depend on nothing; only the activation tiles wait for the previous phase), and that what
// another workgroup will read leaves as write-through stores.  The row phases re-partition add_rmsnorm_kernel (1024 threads per row) and
// qkv_finish_kernel onto 256-thread workgroups with the same per-element arithmetic and the same reduction trees.  Results are bit-identical to the
// launch-per-kernel step (tests/test_gpu_parity.py::test_wide_chain_is_bit_identical, test_big_batch_equals_single).
#include "common.h"
#include "gridbar.h"
#include "kernels.h"

namespace {
using namespace gridbar;

__device__ __forceinline__ int swz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

struct GemmPh {
    const bf16_t* A; int lda;      // [M, K] activations (written by an earlier phase / launch)
    const bf16_t* W;               // decode tile copy [N/16][K/64][64 lanes][16]
    int M, N, K;
    float* part; int ks;           // EPI_PARTIAL: fp32 partial rows [ks][SK_ROWS][N]
    bf16_t* C; int ldc;            // EPI_SWIGLU: [M, N/2]
};

// One 128 x 128 tile (EPI_SWIGLU: all of K; EPI_PARTIAL: one of ks K-chunks) per workgroup blockIdx.x < nitems.  first = no barrier in front (the launch's
// first phase: its activations come from the previous kernel).
template <int EPI>
__device__ __forceinline__ void gemm_phase(char* smem, const GemmPh& p, int nitems, GridBar& gb, bool first, int prefetch, int* s_flag) {
    constexpr int BM = 128, BN = 128, WN = 2, NW = 4, NTHR = 256, NSTAGE = 4;
    constexpr int TM = 4, TN = 4;
    constexpr int A_BYTES = BM * 128, STAGE = 2 * A_BYTES;
    constexpr int A_IT = 4, W_IT = 4, PPT = A_IT + W_IT;
    constexpr bool GLU = (EPI == EPI_SWIGLU);
    constexpr int OSTRIDE = BN * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wid / WN, wn = wid % WN;
    const bool mine = (int)blockIdx.x < nitems && !gb.dead;
    const int ntn = p.N / BN;
    const int item = mine ? (int)blockIdx.x : 0;
    const int kchunk = EPI == EPI_PARTIAL ? item / ntn : 0;
    // (the launch-per-kernel form: blockIdx = chunk * tiles + tile, tile = xcd_remap(...) over the tiles of one chunk; one row panel: tm = 0, tn = t)
    const int t = xcd_remap(EPI == EPI_PARTIAL ? item - kchunk * ntn : item, ntn);
    const int n0 = t * BN;
    const char* asrc[A_IT];
    const char* wsrc[W_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int slot = i * NTHR + tid, row = slot >> 3, kc = (slot & 7) ^ ((row >> 1) & 7);
        const int am = min(row, p.M - 1);
        asrc[i] = reinterpret_cast<const char*>(p.A) + ((size_t)am * p.lda + kc * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int q = i * NW + wid;
        wsrc[i] = reinterpret_cast<const char*>(p.W) + (((size_t)(n0 / 16 + (q >> 1)) * (p.K / 64)) * 1024 + lane * 16 + (q & 1) * 8) * 2;
    }
    auto issue_a = [&](int kt) {
        char* sa = smem + (kt % NSTAGE) * STAGE;
        const int ko = kt * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sa + (i * NTHR + wid * 64) * 16), 16, 0, 0);
    };
    auto issue_w = [&](int kt) {
        char* sw = smem + (kt % NSTAGE) * STAGE + A_BYTES;
        const int ko = kt * 2048;
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + ko),
                                             (__attribute__((address_space(3))) void*)(sw + (i * NTHR + wid * 64) * 16), 16, 0, 0);
    };
    const int nk_all = p.K / 64;
    const int kt0 = EPI == EPI_PARTIAL ? kchunk * (nk_all / p.ks) : 0;
    const int nk = EPI == EPI_PARTIAL ? kt0 + nk_all / p.ks : nk_all;
    const int nkl = nk - kt0;
    // the weight tiles of the ring's first three stages go out before the barrier — from every wave but the one that polls (its poll replies would queue
    // behind them); then the barrier; then that wave's weight pieces and everybody's activation tiles
    const bool early = first || (prefetch && wid != 0);
    if (mine && early) {
#pragma unroll
        for (int d = 0; d < NSTAGE - 1; ++d)
            if (d < nkl) issue_w(kt0 + d);
    }
    if (!first) bar_wait(gb, s_flag);
    if (gb.dead || !mine) return;
    if (!early) {
#pragma unroll
        for (int d = 0; d < NSTAGE - 1; ++d)
            if (d < nkl) issue_w(kt0 + d);
    }
#pragma unroll
    for (int d = 0; d < NSTAGE - 1; ++d)
        if (d < nkl) issue_a(kt0 + d);

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int aoff[TM], woff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = swz(wm * (BM / 2) + i * 16 + r, 2 * g);
#pragma unroll
    for (int j = 0; j < TN; ++j) woff[j] = ((wn * (BN / WN) / 16 + j) * 2) * 1024 + lane * 16;
    constexpr int A_KS = 4, W_KS = 10;

    for (int kt = kt0; kt < nk; ++kt) {
        // tile kt landed.  Issue order of this wave: [W0 W1 W2] [A0 A1 A2] then (A_{j+3} W_{j+3}) per iteration j; loads retire in order, so "at most n
        // younger pieces outstanding" means tile j is complete: j = 0: A1 A2; j = 1: A2 + tile 3; j >= 2: two whole tiles (fewer at the end of the chunk)
        const int j = kt - kt0, left = nkl - 1 - j;
        if (j == 0) {
            if (nkl >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * A_IT) : "memory");
            else if (nkl == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (j == 1) {
            if (nkl >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + PPT) : "memory");
            else if (nkl == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (left >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NSTAGE - 1 < nk) { issue_a(kt + NSTAGE - 1); issue_w(kt + NSTAGE - 1); }
        const char* sa = smem + (kt % NSTAGE) * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t wf[TN], ac[2], an[2];
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) wf[jj] = *reinterpret_cast<const bf16x8_t*>(sw + (woff[jj] ^ (ks << W_KS)));
            ac[0] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[0] ^ (ks << A_KS)));
            ac[1] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[1] ^ (ks << A_KS)));
            __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#pragma unroll
            for (int ip = 0; ip < TM / 2; ++ip) {
                if (ip + 1 < TM / 2) {
                    an[0] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[2 * ip + 2] ^ (ks << A_KS)));
                    an[1] = *reinterpret_cast<const bf16x8_t*>(sa + (aoff[2 * ip + 3] ^ (ks << A_KS)));
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) acc[2 * ip + ii][jj] = mfma16(wf[jj], ac[ii], acc[2 * ip + ii][jj]);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                if (ip + 1 < TM / 2) { ac[0] = an[0]; ac[1] = an[1]; }
            }
        }
    }
    if constexpr (EPI == EPI_PARTIAL) {
        // fp32 accumulators to the chunk's partial rows (write-through: the next phase's workgroups sum them)
        const size_t base = (size_t)kchunk * SK_ROWS * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = wm * (BM / 2) + i * 16 + r;
            if (m < p.M) {
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
                    st_wt16(p.part, (base + (size_t)m * p.N + n0 + wn * (BN / WN) + jj * 16 + g * 4) * 4, __builtin_bit_cast(u32x4_t, acc[i][jj]));
            }
        }
        return;
    } else {
        constexpr int OUTW = BN / 2, CPR = OUTW / 8, OIT = BM * CPR / NTHR;
        const int on0 = n0 / 2;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = wm * (BM / 2) + i * 16 + r;
#pragma unroll
            for (int jj = 0; jj < TN / 2; ++jj) {
                const int nl = wn * (BN / WN / 2) + jj * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gt = acc[i][2 * jj][q], up = acc[i][2 * jj + 1][q];
                    v[q] = gt * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;   // silu(g)*u
                }
                *reinterpret_cast<uint2*>(smem + mrow * OSTRIDE + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < OIT; ++it) {
            const int c = it * NTHR + tid;
            const int row = c / CPR, ch = c - row * CPR;
            if (row >= p.M) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + row * OSTRIDE + ch * 16);
            st_wt16(p.C, ((size_t)row * p.ldc + on0 + ch * 8) * 2, u32x4_t{v.x, v.y, v.z, v.w});
        }
        (void)GLU;
    }
}

// add_rmsnorm_kernel (decode.hip: one 1024-thread workgroup per row, 4 elements per thread) on a 256-thread workgroup: thread (wave w, lane l) plays the
// original's threads (wave 4w + j, lane l), j = 0..3 — the same per-element sums in chunk order, the same 64-lane reduction tree per original wave, the
// same 16-term sum of the wave results.  N <= 4096.  Row b = blockIdx.x < B.
__device__ __forceinline__ void add_rmsnorm_phase(const float* part, int KS, bf16_t* X /* residual in, new residual out */, const bf16_t* w, bf16_t* Y, int B, int N,
                                                  float eps, GridBar& gb, int* s_flag, float* s_red) {
    bar_wait(gb, s_flag);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (gb.dead || b >= B) return;
    float v[4][4];
    uint2 wv[4];
    bool on[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = (wid * 4 + j) * 64 + lane;              // the original thread index: elements 4c .. 4c + 3
        on[j] = c < (N >> 2);
        float ss = 0.f;
        wv[j] = make_uint2(0u, 0u);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = 0.f;
        if (on[j]) {
            wv[j] = *reinterpret_cast<const uint2*>(w + c * 4);
            const uint2 rr = *reinterpret_cast<const uint2*>(X + (size_t)b * N + c * 4);
            const float* p0 = part + (size_t)b * N + c * 4;
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < KS; k0 += 8) {
                f32x4_t t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    t[u] = k0 + u < KS ? *reinterpret_cast<const f32x4_t*>(p0 + (size_t)(k0 + u) * SK_ROWS * N) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) a += t[u];
            }
            const float x0 = bf2f(f2bf(a[0])) + bflo(rr.x), x1 = bf2f(f2bf(a[1])) + bfhi(rr.x);
            const float x2 = bf2f(f2bf(a[2])) + bflo(rr.y), x3 = bf2f(f2bf(a[3])) + bfhi(rr.y);
            const uint2 xo = make_uint2(pack2bf(x0, x1), pack2bf(x2, x3));
            st_wt8(X, ((size_t)b * N + c * 4) * 2, xo.x, xo.y);
            v[j][0] = bflo(xo.x); v[j][1] = bfhi(xo.x); v[j][2] = bflo(xo.y); v[j][3] = bfhi(xo.y);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += v[j][e] * v[j][e];
        }
        ss = wave_sum(ss);
        if (lane == 0) s_red[wid * 4 + j] = ss;
    }
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += s_red[i];
    const float rstd = rsqrtf(tot / (float)N + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (on[j]) {
            const int c = (wid * 4 + j) * 64 + lane;
            const float o0 = v[j][0] * rstd * bflo(wv[j].x), o1 = v[j][1] * rstd * bfhi(wv[j].x);
            const float o2 = v[j][2] * rstd * bflo(wv[j].y), o3 = v[j][3] * rstd * bfhi(wv[j].y);
            st_wt8(Y, ((size_t)b * N + c * 4) * 2, pack2bf(o0, o1), pack2bf(o2, o3));
        }
    }
    __syncthreads();          // s_red is reused by the next row phase
}

// qkv_finish_kernel (decode.hip) over a grid-stride loop: partial rows -> roped bf16 q rows, k / v appended to the next layer's caches
__device__ __forceinline__ void qkv_finish_phase(const DecodeWideArgs& A, GridBar& gb, int* s_flag) {
    bar_wait(gb, s_flag);
    if (gb.dead) return;
    constexpr int HD = 128, HALF = 64;
    const int nq = A.NQ, nkv = A.NKV, nh = nq + 2 * nkv, ldq = A.QKV, ks = A.ks_q;
    const int total = A.B * nh * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int c = idx & 7, hh = (idx >> 3) % nh, b = (idx >> 3) / nh;
        const int p_new = A.pos[b];
        const float* pp = A.part + (size_t)b * ldq + (size_t)hh * HD + c * 8;
        f32x4_t x[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int k = 0; k < ks; ++k) {
            const float* pk = pp + (size_t)k * SK_ROWS * ldq;
            x[0] += *reinterpret_cast<const f32x4_t*>(pk);
            x[1] += *reinterpret_cast<const f32x4_t*>(pk + 4);
            x[2] += *reinterpret_cast<const f32x4_t*>(pk + HALF);
            x[3] += *reinterpret_cast<const f32x4_t*>(pk + HALF + 4);
        }
        uint32_t a[4] = {pack2bf(x[0][0], x[0][1]), pack2bf(x[0][2], x[0][3]), pack2bf(x[1][0], x[1][1]), pack2bf(x[1][2], x[1][3])};
        uint32_t bb[4] = {pack2bf(x[2][0], x[2][1]), pack2bf(x[2][2], x[2][3]), pack2bf(x[3][0], x[3][1]), pack2bf(x[3][2], x[3][3])};
        if (hh >= nq + nkv) {                                // v: down a column of V^T (read by the next kernel: plain stores)
            bf16_t* vb = A.vc_next + (size_t)A.slots[b] * A.slot_stride + (size_t)(hh - nq - nkv) * A.kv_head_stride + p_new;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vb[(size_t)(c * 8 + 2 * e) * A.ctx_stride] = (bf16_t)(a[e] & 0xffffu);
                vb[(size_t)(c * 8 + 2 * e + 1) * A.ctx_stride] = (bf16_t)(a[e] >> 16);
                vb[(size_t)(HALF + c * 8 + 2 * e) * A.ctx_stride] = (bf16_t)(bb[e] & 0xffffu);
                vb[(size_t)(HALF + c * 8 + 2 * e + 1) * A.ctx_stride] = (bf16_t)(bb[e] >> 16);
            }
            continue;
        }
        const float4* cp = reinterpret_cast<const float4*>(A.cos_t + (size_t)p_new * HALF + c * 8);
        const float4* sp = reinterpret_cast<const float4*>(A.sin_t + (size_t)p_new * HALF + c * 8);
        const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        uint32_t o1[4], o2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x1l = bflo(a[e]), x1h = bfhi(a[e]), x2l = bflo(bb[e]), x2h = bfhi(bb[e]);
            o1[e] = pack2bf(x1l * cs[2 * e] - x2l * sn[2 * e], x1h * cs[2 * e + 1] - x2h * sn[2 * e + 1]);
            o2[e] = pack2bf(x2l * cs[2 * e] + x1l * sn[2 * e], x2h * cs[2 * e + 1] + x1h * sn[2 * e + 1]);
        }
        bf16_t* dst = hh < nq ? A.dQKV + (size_t)b * ldq + (size_t)hh * HD
                              : A.kc_next + (size_t)A.slots[b] * A.slot_stride + (size_t)(hh - nq) * A.kv_head_stride + (size_t)p_new * HD;
        *reinterpret_cast<uint4*>(dst + c * 8) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<uint4*>(dst + HALF + c * 8) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    }
}

__global__ __launch_bounds__(256) void decode_wide_chain_kernel(DecodeWideArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_flag;
    __shared__ float s_red[16];
    GridBar gb{A.bar, A.err, 0u, (int)gridDim.x, false, 0};
    bar_begin(gb, &s_flag);
    const unsigned final_epoch = gb.epoch + (A.last ? 4u : 6u);          // bar_arrive calls of this launch
    const int H = A.H, I = A.I, B = A.B;
    {   // o-proj: dO [B, H] . Wo^T -> partial rows
        const GemmPh p{A.dO, H, A.wo, B, H, H, A.part, A.ks_o, nullptr, 0};
        gemm_phase<EPI_PARTIAL>(smem, p, (H / 128) * A.ks_o, gb, true, A.prefetch, &s_flag);
        bar_arrive(gb);
    }
    add_rmsnorm_phase(A.part, A.ks_o, A.dX, A.rms2, A.dH, B, H, A.eps, gb, &s_flag, s_red);
    bar_arrive(gb);
    {   // gate|up + SwiGLU: dH . Wgu^T -> dACT [B, I]
        const GemmPh p{A.dH, H, A.wgu, B, 2 * I, H, nullptr, 1, A.dACT, I};
        gemm_phase<EPI_SWIGLU>(smem, p, (2 * I) / 128, gb, false, A.prefetch, &s_flag);
        bar_arrive(gb);
    }
    {   // down: dACT . Wd^T -> partial rows
        const GemmPh p{A.dACT, I, A.wd, B, H, I, A.part, A.ks_d, nullptr, 0};
        gemm_phase<EPI_PARTIAL>(smem, p, (H / 128) * A.ks_d, gb, false, A.prefetch, &s_flag);
        bar_arrive(gb);
    }
    add_rmsnorm_phase(A.part, A.ks_d, A.dX, A.rms_next, A.dH, B, H, A.eps, gb, &s_flag, s_red);
    if (A.last) { bar_end(gb, final_epoch); return; }          // (past the launch's last bar_wait)
    bar_arrive(gb);
    {   // the next layer's qkv projection: dH . Wqkv^T -> partial rows
        const GemmPh p{A.dH, H, A.wqkv_next, B, A.QKV, H, A.part, A.ks_q, nullptr, 0};
        gemm_phase<EPI_PARTIAL>(smem, p, (A.QKV / 128) * A.ks_q, gb, false, A.prefetch, &s_flag);
        bar_arrive(gb);
    }
    qkv_finish_phase(A, gb, &s_flag);
    bar_end(gb, final_epoch);
}

}  // namespace

size_t decode_wide_bar_bytes() { return gridbar::bar_bytes(); }

int launch_decode_wide_chain(const DecodeWideArgs& a, hipStream_t s) {
    const int ncu = decode_b1_num_cus();
    if (ncu <= 0) return TRACE_ERR_HIP;
    if (a.B < 1 || a.B > 128 || a.H % 128 || a.QKV % 128 || (2 * a.I) % 128 || a.H > 4096 || a.ks_o < 1 || a.ks_d < 1 || a.ks_q < 1) return TRACE_ERR_STATE;
    if ((a.H / 64) % a.ks_o || (a.I / 64) % a.ks_d || (a.H / 64) % a.ks_q) return TRACE_ERR_STATE;
    // every phase one item per workgroup, all workgroups resident (one per CU: 128 KB of LDS each)
    const int items[4] = {(a.H / 128) * a.ks_o, (2 * a.I) / 128, (a.H / 128) * a.ks_d, (a.QKV / 128) * a.ks_q};
    for (int i = 0; i < 4; ++i)
        if (items[i] > ncu) return TRACE_ERR_STATE;
    if (a.B > ncu || !a.bar || !a.err) return TRACE_ERR_STATE;
    constexpr int LDSB = 4 * 2 * 128 * 128;
    static LdsGrant grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(decode_wide_chain_kernel), LDSB)) return TRACE_ERR_HIP;
    hipLaunchKernelGGL(decode_wide_chain_kernel, dim3(ncu), dim3(256), LDSB, s, a);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
