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
// Round 2 (ISA read with tools/isa_mix.py): the per-element validity test of partial / diagonal tiles had been if-converted
// by the compiler into ~330 compare/select instructions executed on EVERY key tile (of ~490 VALU per tile against 16 MFMAs:
// the kernel was VALU-issue bound at 0.21 of the MFMA peak).  The test now sits behind a wave-uniform branch (readfirstlane)
// and only edge tiles run the masked instantiation of the tile body; a short tail of keys (577 = 9 x 64 + 1 in the ViT) never
// becomes a tile at all: those keys initialise the online-softmax state on the VALU (dot product, exp2, one scaled V row)
// before the tile loop.  V^T rows are stored with the two 4-key groups a lane needs side by side, so each PV operand is one
// ds_read_b128 instead of two ds_read_b64.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BKV = 64;
constexpr int VROW = 144;   // bytes per V^T row in LDS: 64 kv * 2 + 16 pad (odd multiple of 16: conflict-free ds_read_b128)
constexpr int TAILV = 8;    // at most this many trailing keys (nkv % 64) are folded in on the VALU instead of a masked tile

template <int HD>
__device__ __forceinline__ int kswz(int row, int kc) {
    return row * (HD * 2) + ((kc ^ (HD == 64 ? ((row >> 1) & 7) : (row & 15))) << 4);
}

// one key tile: S^T = K.Q^T, online softmax (MASK: per-element validity, edge tiles only), O^T += V^T.P
// (a function template with explicit references, not a by-reference lambda: with those hipcc kept the captured arrays in scratch)
template <int HD, bool MASK>
__device__ __forceinline__ void attn_tile(const char* kb, const char* vb, int kv0, const bf16x8_t (&qf)[HD / 16], f32x16_t (&oacc)[HD / 32],
                                          float& m, float& l, float sc, int c32, int h, int nkv_main, int causal, int qabs, int off) {
    f32x16_t S[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[st][r] = 0.f;
        const int row = st * 32 + c32;
#pragma unroll
        for (int s_ = 0; s_ < HD / 16; ++s_) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + kswz<HD>(row, s_ * 2 + h));
            S[st] = mfma32(kf, qf[s_], S[st]);
        }
    }
    if (MASK) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool ok = kv < nkv_main && (!causal || kv <= qabs + off);
                S[st][r] = ok ? S[st][r] : -1e30f;
            }
    }
    // ---- online softmax (base 2; the score scale is folded into the exp2 argument: p = 2^(s*sc - m)) ----
    float mloc = -1e30f;
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
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vb + (ht * 32 + c32) * VROW + ks * 32 + h * 16);
            oacc[ht] = mfma32(vf, pf.v, oacc[ht]);
        }
    }
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
    // 1-D grid, XCD-aware: the dispatcher places block i on XCD i % 8 and the 8 XCDs do not share an L2, so the blocks that read
    // the same K/V (the 5 query blocks of one ViT (frame, head); every query tile of one prefill kv head) must be NEIGHBOURS IN THE
    // SAME XCD'S SEQUENCE — as (x, y, z) grid they were spread over 5-8 XCDs and each L2 fetched its own copy from HBM (the ViT
    // launch read its 400 MB of K/V five times: 4.4 TB/s, i.e. it ran at the HBM roofline, not the MFMA one).
    const int nqt = (a.nq_rows + 31) / 32;
    const int nqb = GQA ? nqt : (nqt + 3) / 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int b, kvh, qblk;
    if (GQA) {          // [kv head][batch][query tile], heaviest (last) causal tiles first: with 8 kv heads, XCD x = kv head x
        qblk = nqb - 1 - lid % nqb;
        b = (lid / nqb) % a.batch;
        kvh = lid / (nqb * a.batch);
    } else {            // [batch][head][query block]
        qblk = lid % nqb;
        kvh = (lid / nqb) % a.kv_heads;
        b = lid / (nqb * a.kv_heads);
    }
    const int qh = GQA ? kvh * (a.heads / a.kv_heads) + wid : kvh;
    const int qt = GQA ? qblk : qblk * 4 + wid;
    const int q0 = qt * 32;
    const bool active = q0 < a.nq_rows;
    const int off = a.nkv_rows - a.nq_rows;
    const int qabs = q0 + c32;

    // trailing keys (nkv % 64 of them, when few) are folded in on the VALU: the tile loop covers whole tiles only
    const int rem = a.nkv_rows % BKV;
    const bool vtail = !GQA && a.Vrow != nullptr && !a.causal && rem > 0 && rem <= TAILV && a.nkv_rows > BKV;
    const int nkv_main = vtail ? a.nkv_rows - rem : a.nkv_rows;
    // block-level kv extent
    int kv_limit = nkv_main;
    if (a.causal) {
        const int qmax = (GQA ? q0 : qblk * 128 + 96) + 31 + off;   // last query row of the block
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
            char* vp_ = vb_ + vrow[i] * VROW + (vkc[i] >> 1) * 32 + (vkc[i] & 1) * 8;                          \
            *reinterpret_cast<u32x2*>(vp_) = u32x2{rv[i][0], rv[i][1]};                                     \
            *reinterpret_cast<u32x2*>(vp_ + 16) = u32x2{rv[i][2], rv[i][3]};                                \
        }                                                                                                      \
    }

    f32x16_t oacc[HD / 32];
#pragma unroll
    for (int i = 0; i < HD / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const float sc = a.scale * 1.4426950408889634f;

    // ---- trailing keys on the VALU (non-causal only): state after them = softmax over those keys alone ----
    if (vtail && active) {
        for (int j = 0; j < rem; ++j) {
            const int kv = nkv_main + j;
            const bf16_t* kp = kbase + (size_t)kv * a.k_rs + h * 8;
            float dot = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < HD / 16; ++s_) {
                const uint4 kk = *reinterpret_cast<const uint4*>(kp + s_ * 16);
                union { bf16x8_t v; uint32_t u[4]; } qq;
                qq.v = qf[s_];
                const uint32_t ku[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dot = fmaf(bflo(qq.u[e]), bflo(ku[e]), dot);
                    dot = fmaf(bfhi(qq.u[e]), bfhi(ku[e]), dot);
                }
            }
            dot += __shfl_xor(dot, 32, 64);
            const float mnew = fmaxf(m, dot * sc);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            const float p = __builtin_amdgcn_exp2f(fmaf(dot, sc, -mnew));
            m = mnew;
            l = l * alpha + (h == 0 ? p : 0.f);            // the epilogue adds the two halves' sums
            const bf16_t* vp = a.Vrow + (size_t)b * a.vr_bs + (size_t)kvh * a.vr_hs + (size_t)kv * a.vr_rs + 4 * h;
#pragma unroll
            for (int ht = 0; ht < HD / 32; ++ht)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const uint2 vv = *reinterpret_cast<const uint2*>(vp + ht * 32 + 8 * rg);
                    oacc[ht][rg * 4 + 0] = fmaf(p, bflo(vv.x), oacc[ht][rg * 4 + 0] * alpha);
                    oacc[ht][rg * 4 + 1] = fmaf(p, bfhi(vv.x), oacc[ht][rg * 4 + 1] * alpha);
                    oacc[ht][rg * 4 + 2] = fmaf(p, bflo(vv.y), oacc[ht][rg * 4 + 2] * alpha);
                    oacc[ht][rg * 4 + 3] = fmaf(p, bfhi(vv.y), oacc[ht][rg * 4 + 3] * alpha);
                }
        }
    }

    if (nt > 0) {
        ATTN_GLOAD(0)
        ATTN_LSTORE(0)
    }
    __syncthreads();

    // tiles [0, nt_plain) need no per-element test (all 64 keys exist and, when causal, are visible to every row of the block);
    // the few edge tiles run the masked body in a loop of their own — one body per loop keeps both free of spills, and the
    // split is block-uniform so every wave passes the same barriers
    int nt_plain = nkv_main / BKV;
    if (a.causal) nt_plain = min(nt_plain, max(0, ((GQA ? q0 : qblk * 128) + off + 1) / BKV));
    nt_plain = min(nt_plain, nt);
#define ATTN_LOOP(T0_, T1_, MASK_)                                                                             \
    for (int t = (T0_); t < (T1_); ++t) {                                                                      \
        const bool more = t + 1 < nt && a.dbg != 1;                                                            \
        if (more) ATTN_GLOAD(t + 1)                                                                            \
        const char* kb = smem + (t & 1) * BUF;                                                                 \
        if (active && a.dbg != 2) attn_tile<HD, MASK_>(kb, kb + KT_BYTES, t * BKV, qf, oacc, m, l, sc, c32, h, nkv_main, a.causal, qabs, off); \
        if (more) ATTN_LSTORE((t + 1) & 1)                                                                     \
        __syncthreads();                                                                                       \
    }
    ATTN_LOOP(0, nt_plain, false)
    ATTN_LOOP(nt_plain, nt, true)
#undef ATTN_LOOP

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

// =====================================================================================================================
// ViT attention, round 2: head_dim 64, non-causal, one launch = frames x heads x 577 tokens.
// Knock-out runs of the kernel above (round 2's tools/attn_vit_probe.py, 170 frames; today tools/attn_vit_big_probe.py): tile math alone 387 us, K/V staging alone (global
// loads -> registers -> LDS stores -> barrier) 261 us, together 466-510 us, i.e. 0.2 of the MFMA peak with the matrix pipe 27 % busy:
// the loop was bound by VALU issue (135 plain + 33 transcendental instructions per 16 MFMAs) and by the staging instructions.
// This kernel removes most of both:
//  * K and V^T tiles arrive by LDS-DMA (global_load_lds, 16 B per lane, no registers, no ds_write) into a 3-stage ring, two
//    tiles ahead, behind a counted vmcnt + one raw s_barrier per tile.  K rows are XOR-swizzled on the SOURCE address; V^T comes
//    from transpose_v's PERMUTED layout (within every 16 keys the two 4-key groups a lane needs are adjacent) so each PV operand
//    is one swizzled ds_read_b128.
//  * softmax per score element is exp2 + (1/2) max3 + (1/2) cvt_pk + (1/2) pk_add: the score scale and log2(e) are folded into
//    the Q fragments once per block, and the running maximum is folded into the QK^T MFMA itself — its accumulator starts at
//    -m instead of 0 — so the MFMA result is already the exp2 argument.  The maximum moves lazily: O, l and m are rescaled only
//    when some row's tile maximum exceeds the current reference by more than RESCALE_THR (exact arithmetic either way: p is
//    exponentiated after the decision, so nothing is ever rescaled twice or not at all); p <= 2^RESCALE_THR fits bf16's exponent range.
//  * the 577th key (nkv % 64 of them) initialises the state on the VALU as in the kernel above.
constexpr int DSTAGE = 16384, DSTAGES = 3;
constexpr float RESCALE_THR = 6.f;

__device__ __forceinline__ void glds16(const bf16_t* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// VROW = true (round 3): V is taken ROW-MAJOR straight from the QKV buffer — no transpose_v pass (64 us per layer at 170 frames, a pure
// mover) and no V^T scratch.  The V half of a stage is then an image [d-half ht][key 0..63][32 d] (64-byte rows; a 1 KB LDS-DMA piece = 16
// keys x 64 B, lane = (key, 16-byte chunk)), and a PV operand — 8 keys of one d per lane — is two ds_read_b64_tr_b16 (gfx950's LDS
// transpose read: the 16 lanes of a group fetch a [4 keys][16 d] block, 8 bytes each, and lane i receives column i): group (cg, h) of
// the wave reads keys 16 ks + 4h .. + 3 (and + 8) x d ht*32 + cg*16 .. + 15, i.e. the 32 lanes served together read 4 consecutive
// 64-byte rows = every bank once.  The k-slot <-> key map is the one the S accumulators dictate, as before.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
template <bool VROW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_vit_dma_kernel(AttnArgs a) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    const int nqt = (a.nq_rows + 31) / 32, nqb = (nqt + 3) / 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % nqb, kvh = (lid / nqb) % a.kv_heads, b = lid / (nqb * a.kv_heads);
    const int q0 = (qblk * 4 + wid) * 32;
    const bool active = q0 < a.nq_rows;
    const int qabs = q0 + c32;
    const int rem = a.nkv_rows % BKV, nkv_main = a.nkv_rows - rem, nt = nkv_main / BKV;
    const bf16_t* kbase = a.K + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
    const bf16_t* vbase = VROW ? a.Vrow + (size_t)b * a.vr_bs + (size_t)kvh * a.vr_hs : a.V + (size_t)b * a.v_bs + (size_t)kvh * a.v_hs;

    // ---- LDS-DMA: per tile the block moves 8 K pieces + 8 V pieces of 1 KB (K, V^T: 8 rows x 128 B; row-major V: 16 keys x 64 B of one
    //      d-half); wave w issues pieces 2w, 2w+1 of each
    const bf16_t* ksrc[2];
    const bf16_t* vsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (2 * wid + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        ksrc[j] = kbase + (size_t)r * a.k_rs + c * 8;
        if (VROW) {
            const int pi = 2 * wid + j;                  // piece (ht = pi >> 2, keys (pi & 3) * 16 ..): lane = (key, chunk)
            vsrc[j] = vbase + (size_t)((pi & 3) * 16 + (lane >> 2)) * a.vr_rs + (pi >> 2) * 32 + (lane & 3) * 8;
        } else vsrc[j] = vbase + (size_t)r * a.v_rs + c * 8;
    }
    const long kstep = (long)BKV * a.k_rs;
    const long vstep = VROW ? (long)BKV * a.vr_rs : (long)BKV;
    auto issue = [&](int t, int stage) {
        char* st = smem + stage * DSTAGE + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16(ksrc[j] + (size_t)t * kstep, st + j * 1024);
            glds16(vsrc[j] + (size_t)t * vstep, st + 8192 + j * 1024);
        }
    };
    // row-major V: this lane's byte offset inside a stage's V image for (ks = 0, ht = 0, first key quad): key 4h + (i >> 2), d cg*16 + (i & 3)*4
    const int vtr_off = ((4 * h + ((lane & 15) >> 2)) * 64) + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    issue(0, 0);
    if (nt > 1) issue(1, 1);

    // ---- Q fragments, scaled once: q' = bf16(q * scale * log2 e), so that K.q' is the exp2 argument
    const float sc = a.scale * 1.4426950408889634f;
    bf16x8_t qf[HD / 16];
    {
        const int qr = min(qabs, a.nq_rows - 1);
        const bf16_t* qp = a.Q + (size_t)b * a.q_bs + (size_t)kvh * a.q_hs + (size_t)qr * a.q_rs + h * 8;
#pragma unroll
        for (int s_ = 0; s_ < HD / 16; ++s_) {
            const uint4 q4 = *reinterpret_cast<const uint4*>(qp + s_ * 16);
            union { bf16x8_t v; uint32_t u[4]; } o;
            o.u[0] = pack2bf(bflo(q4.x) * sc, bfhi(q4.x) * sc);
            o.u[1] = pack2bf(bflo(q4.y) * sc, bfhi(q4.y) * sc);
            o.u[2] = pack2bf(bflo(q4.z) * sc, bfhi(q4.z) * sc);
            o.u[3] = pack2bf(bflo(q4.w) * sc, bfhi(q4.w) * sc);
            qf[s_] = o.v;
        }
    }

    f32x16_t oacc[HD / 32];
#pragma unroll
    for (int i = 0; i < HD / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = 0.f, l = 0.f;        // m: the reference every exponent is taken against (exp2 domain); meaningless while `first`
    bool first = true;

    // ---- trailing keys on the VALU
    if (active) {
        for (int j = 0; j < rem; ++j) {
            const int kv = nkv_main + j;
            const bf16_t* kp = kbase + (size_t)kv * a.k_rs + h * 8;
            float dot = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < HD / 16; ++s_) {
                const uint4 kk = *reinterpret_cast<const uint4*>(kp + s_ * 16);
                union { bf16x8_t v; uint32_t u[4]; } qq;
                qq.v = qf[s_];
                const uint32_t ku[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dot = fmaf(bflo(qq.u[e]), bflo(ku[e]), dot);
                    dot = fmaf(bfhi(qq.u[e]), bfhi(ku[e]), dot);
                }
            }
            dot += __shfl_xor(dot, 32, 64);
            const float mnew = first ? dot : fmaxf(m, dot);
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m - mnew);
            const float p = __builtin_amdgcn_exp2f(dot - mnew);
            m = mnew; first = false;
            l = l * alpha + (h == 0 ? p : 0.f);
            const bf16_t* vp = a.Vrow + (size_t)b * a.vr_bs + (size_t)kvh * a.vr_hs + (size_t)kv * a.vr_rs + 4 * h;
#pragma unroll
            for (int ht = 0; ht < HD / 32; ++ht)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const uint2 vv = *reinterpret_cast<const uint2*>(vp + ht * 32 + 8 * rg);
                    oacc[ht][rg * 4 + 0] = fmaf(p, bflo(vv.x), oacc[ht][rg * 4 + 0] * alpha);
                    oacc[ht][rg * 4 + 1] = fmaf(p, bfhi(vv.x), oacc[ht][rg * 4 + 1] * alpha);
                    oacc[ht][rg * 4 + 2] = fmaf(p, bflo(vv.y), oacc[ht][rg * 4 + 2] * alpha);
                    oacc[ht][rg * 4 + 3] = fmaf(p, bfhi(vv.y), oacc[ht][rg * 4 + 3] * alpha);
                }
        }
    }
    f32x16_t cinit;                                  // accumulator start of the QK^T MFMAs: -m in every slot (0 while `first`)
    {
        const float ci = first ? 0.f : -m;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = ci;
    }

    int stage = 0, stage2 = 2 % DSTAGES;
    for (int t = 0; t < nt; ++t) {
        // tile t landed: this wave's pieces by its own counted vmcnt (the 4 pieces of tile t+1 may stay in flight), everybody's by the
        // barrier; the barrier also says every wave is done reading tile t-1, whose stage the DMA issued next overwrites
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < nt && a.dbg != 1) issue(t + 2, stage2);
        if (active && a.dbg != 2) {
            const char* kb = smem + stage * DSTAGE;
            const char* vb = kb + 8192;
            f32x16_t S[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int row = st * 32 + c32;
#pragma unroll
                for (int s_ = 0; s_ < HD / 16; ++s_) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + kswz<HD>(row, s_ * 2 + h));
                    if (s_ == 0) {
                        // D != C: the builtin ties the result to its accumulator operand, which made the compiler copy cinit into S
                        // first (27 v_mov per tile); the instruction itself takes separate registers (early-clobber: D must not
                        // overlap the sources).  The s_nop covers the VALU-write -> MFMA-read wait states hipcc does not pad inside asm.
                        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" TRACE_EL " %0, %1, %2, %3" : "=&v"(S[st]) : "v"(kf), "v"(qf[0]), "v"(cinit));
                    } else {
                        S[st] = mfma32(kf, qf[s_], S[st]);
                    }
                }
            }
            // tile maximum relative to m (S already is s - m)
            float mt = fmaxf(S[0][0], S[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, S[0][r]), S[1][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const bool need = first || mt > RESCALE_THR;
            if (__any(need)) {                                   // rare after the first tile: move the reference up for the rows that need it
                const float d = need ? mt : 0.f;
                const float f = first ? 0.f : __builtin_amdgcn_exp2f(-d);       // (first tile: O = l = 0, and 2^-d may overflow)
                m += d;
                l *= f;
#pragma unroll
                for (int i = 0; i < HD / 32; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= f;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[st][r] -= d;
#pragma unroll
                for (int r = 0; r < 16; ++r) cinit[r] = -m;
                first = false;
            }
            float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(S[st][r]), p1 = __builtin_amdgcn_exp2f(S[st][r + 1]);
                    ls0 += p0; ls1 += p1;
                    S[st][r] = p0; S[st][r + 1] = p1;
                }
            l += ls0 + ls1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int st = ks >> 1, rb = (ks & 1) * 8;
                union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
                for (int i = 0; i < 4; ++i) pf.u[i] = pack2bf(S[st][rb + 2 * i], S[st][rb + 2 * i + 1]);
#pragma unroll
                for (int ht = 0; ht < HD / 32; ++ht) {
                    bf16x8_t vf;
                    if (VROW) {
                        typedef s16x4_t __attribute__((address_space(3))) * lds_s4;
                        const char* vp = vb + vtr_off + ht * 4096 + ks * 1024;
                        union { bf16x8_t v; s16x4_t q[2]; } u;
                        u.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vp));
                        u.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vp + 512));
                        vf = u.v;
                    } else vf = *reinterpret_cast<const bf16x8_t*>(vb + kswz<HD>(ht * 32 + c32, ks * 2 + h));
                    oacc[ht] = mfma32(vf, pf.v, oacc[ht]);
                }
            }
        }
        stage = stage + 1 == DSTAGES ? 0 : stage + 1;
        stage2 = stage2 + 1 == DSTAGES ? 0 : stage2 + 1;
    }

    if (active && qabs < a.nq_rows) {
        const float lt = l + __shfl_xor(l, 32, 64);
        const float inv = 1.f / lt;
        bf16_t* op = a.O + (size_t)b * a.o_bs + (size_t)kvh * a.o_hs + (size_t)qabs * a.o_rs;
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

// =====================================================================================================================
// ViT attention, round 5: 192 query rows per workgroup on the 16x16x32 MFMA ("big" kernel).
// What the 4 x 32-row kernel above leaves on the table at 577 tokens (rocprofv3, 170 frames x 16 heads: 381 us = 0.24 of the MFMA peak):
//   * 577 = 18 x 32 + 1: the 19th query tile computes 32 rows for ONE, and 19 tiles fill 5 workgroups of 4 waves: 20 wave slots for 18.03 tiles
//     of work (10 %);
//   * every workgroup streams its head's whole K / V (148 KB) from L2 for 128 query rows — 2.0 GB per launch, and the staging alone costs
//     ~250 us of a launch (knock-out runs, profiles/r02_attn_vit_probe.txt);
//   * every K / V fragment read from LDS feeds ONE MFMA; the VALU is the busier pipe (62 % vs 33 % of the SIMD cycles, r02_attn_vit_pmc.txt).
// Here a wave owns 48 query rows = three 16-row tiles of the 16x16x32 MFMA, a workgroup 4 x 48 = 192 rows, and 576 = 3 x 192: three whole
// workgroups per (frame, head), no padded rows, K / V streamed 3 times instead of 5 (1.2 GB), every LDS fragment read feeds three MFMAs, and
// two waves per SIMD (192 registers) with three independent softmax streams each.  Layout, per 64-key tile (both matmuls "swapped", as above):
//   S^T[kv 16][q 16] = K . Q^T   A = K fragment (lane (r, g): key r, d = ks 32 + 8 g ..: one swizzled ds_read_b128), B = Q fragment (registers);
//                                the accumulator starts at -m (the softmax reference), so the result IS the exp2 argument; lane (q, g) holds
//                                the scores of query q against keys 4 g .. 4 g + 3 of the 16-key tile;
//   O^T[d 16][q 16] += V^T . P   B = P: the k-slots (g, 0..3 | 4..7) of lane (q, g) are its own scores of key tiles 2 kp and 2 kp + 1 — P never
//                                leaves the lane; A = V^T fragment with the matching key order: two ds_read_b64_tr_b16 of the row-major V image
//                                (a 16-lane group fetches keys 4 g .. + 3 x 16 d and lane r receives column r).  The V image has 64-byte rows
//                                [d-half][key][32 d] whose two 32-byte halves are swapped for odd (key >> 2) — on the DMA's SOURCE address — so
//                                that the 32 lanes served together (g = 0, 1: 8 keys x 32 bytes) touch every bank once;
//   l[q]            += 1 . P     the row sums ride on the matrix pipe too: A = a constant fragment of ones (no LDS read, 6 of 54 MFMAs per tile)
//                                instead of 48 VALU adds per lane and tile — the sums are then the sums of the ROUNDED p, i.e. exactly the
//                                weights O was accumulated with.
// The softmax reference moves lazily (THR = 8: p <= 256), decided per query row but only inside a wave-uniform slow path (the common tile costs a
// local max + one compare per query tile).  The query rows that 192 does not divide (row 576 of the ViT) go to one extra workgroup per four
// (frame, head) pairs, one wave per pair, on the VALU (a 577-key GEMV: ~2 us per wave, hidden among the MFMA workgroups that read the same K / V).
constexpr int BSTAGE = 16384;
constexpr int BIG_ROWS = 192, BIG_QW = 48, BIG_MAXKV = 1024, BIG_MAXLEFT = 4;
constexpr float BIG_THR = 8.f;

__device__ __forceinline__ void mfma16_from(f32x4_t& d, const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
    // D != C: the builtin ties the result to its accumulator operand (a copy of the -m vector per score tile); early-clobber: D overlaps no source
    // (the s_nop covers the VALU-write -> MFMA-read wait states hipcc does not pad inside asm, as in attn_vit_dma_kernel)
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_" TRACE_EL " %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}

// "Does any of the 16 scores a lane holds exceed the threshold?" on the scores' BIT PATTERNS with integer max3: fmaxf() makes the compiler quiet
// possible signalling NaNs first (one v_max_f32 x, x, x per MFMA result: 60 instructions per key tile for 48 comparisons), and a v_max3_f32 in inline asm
// is invisible to hipcc's hazard recognizer — it was scheduled straight behind the MFMA that writes its operand, inside that MFMA's result latency, read
// stale registers now and then, and two runs of the first hardware test differed in a few bits.  As signed integers, IEEE floats above a positive
// threshold compare like the floats (negative scores are negative integers, NaNs land above everything -> the slow path), which is all the fast path
// needs; the slow path takes its exact maxima with fmaxf.
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int bits_max8(const f32x4_t& a, const f32x4_t& b) {
    int r = imax3(__float_as_int(a[0]), __float_as_int(a[1]), __float_as_int(a[2]));
    r = imax3(r, __float_as_int(a[3]), __float_as_int(b[0]));
    r = imax3(r, __float_as_int(b[1]), __float_as_int(b[2]));
    return max(r, __float_as_int(b[3]));
}
// LDS transpose reads as inline asm.  Through the builtin, hipcc's wait-count pass cannot tell the read from the LDS-DMA writes in flight and puts an
// s_waitcnt vmcnt(0) in front of the first one of every tile — i.e. the wave waits for ALL the tiles it has prefetched before it may read the
// one that landed long ago (the 4 x 32-row kernel above pays exactly that; found in its ISA in round 5).  As asm the reads are opaque to that pass;
// their results are handed to the MFMAs through tr_wait(), an s_waitcnt lgkmcnt(0) that carries the registers as tied operands, so no use (and no
// copy) of a result can be scheduled before the wait.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
__device__ __forceinline__ void tr_read2(u32x2_t& lo, u32x2_t& hi, uint32_t lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(lds_addr) : "memory");
}
__device__ __forceinline__ void tr_wait(u32x2_t (&v)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[3][0]), "+v"(v[3][1])
                 :: "memory");
}

// v_permlane16_swap: rows (16 lanes) 1 and 3 of a <-> rows 0 and 2 of b
__device__ __forceinline__ void swap16_rows(uint32_t& a, uint32_t& b) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    const u2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

template <int NST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_vit_big_kernel(AttnArgs a) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int nqb = a.nq_rows / BIG_ROWS, nleft = a.nq_rows - nqb * BIG_ROWS;
    const int per = 4 * nqb + (nleft > 0 ? 1 : 0);              // workgroups per quad of (frame, head) pairs: 4 x nqb MFMA workgroups (+ one for the leftover rows)
    const int BH = a.batch * a.kv_heads;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int quad = lid / per, rq = lid - quad * per;
    const float sc = a.scale * 1.4426950408889634f;

    if (rq == 4 * nqb) {
        // ---------------- leftover query rows of four (frame, head) pairs: one wave per pair, VALU only, no workgroup barrier ----------------
        const int bh = quad * 4 + wid;
        if (bh >= BH) return;
        const int b = bh / a.kv_heads, kvh = bh - b * a.kv_heads;
        const bf16_t* kbase = a.K + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
        const bf16_t* vbase = a.Vrow + (size_t)b * a.vr_bs + (size_t)kvh * a.vr_hs;
        float* pl = reinterpret_cast<float*>(smem + wid * (BIG_MAXKV * 4));
        const int nit = (a.nkv_rows + 63) >> 6;
        // (First version: lane = d with one 2-byte load per key and lane — 577 dependent load round trips per wave, ~100 us; those workgroups then held
        // ~30 % of the CU slots of the launch.  Now every load is 16 bytes and eight of them are in flight per lane.)
        for (int rr = 0; rr < nleft; ++rr) {
            const int row = nqb * BIG_ROWS + rr;
            const bf16_t* qp = a.Q + (size_t)b * a.q_bs + (size_t)kvh * a.q_hs + (size_t)row * a.q_rs;
            uint4 qv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) qv[c] = *reinterpret_cast<const uint4*>(qp + c * 8);
            // scores of keys it * 64 + lane (exp2 domain): the lane's own 128-byte K row, 8 loads in flight
            float mx = -3.0e38f;
            for (int it = 0; it < nit; ++it) {
                const int key = it * 64 + lane;
                const bf16_t* kp = kbase + (size_t)min(key, a.nkv_rows - 1) * a.k_rs;
                uint4 kk[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) kk[c] = *reinterpret_cast<const uint4*>(kp + c * 8);
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    dot = dot2_el(qv[c].x, kk[c].x, dot); dot = dot2_el(qv[c].y, kk[c].y, dot);
                    dot = dot2_el(qv[c].z, kk[c].z, dot); dot = dot2_el(qv[c].w, kk[c].w, dot);
                }
                dot *= sc;
                if (key < a.nkv_rows) { mx = fmaxf(mx, dot); pl[key] = dot; }
            }
            mx = wave_max(mx);
            float ls = 0.f;
            for (int it = 0; it < nit; ++it) {
                const int key = it * 64 + lane;
                if (key < a.nkv_rows) {
                    const float p = __builtin_amdgcn_exp2f(pl[key] - mx);      // (each lane re-reads what it wrote)
                    pl[key] = p;
                    ls += p;
                }
            }
            ls = wave_sum(ls);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's LDS writes precede its reads below (one in-order LDS queue per wave)
            // O[d] = sum_key p[key] V[key][d]: lane (kg = lane >> 3, c = lane & 7) takes keys kg, kg + 8, ... and d = 8 c .. 8 c + 7 (16-byte loads: a wave
            // instruction covers 8 whole V rows), eight keys per lane in flight; the eight key groups are summed across lanes at the end (fixed order)
            const int kg = lane >> 3, c8 = lane & 7;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            const bf16_t* vp = vbase + c8 * 8;
            for (int k0 = 0; k0 < a.nkv_rows; k0 += 64) {
                uint4 vv[8];
                float pp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int key = k0 + u * 8 + kg;
                    const int kc = min(key, a.nkv_rows - 1);
                    vv[u] = *reinterpret_cast<const uint4*>(vp + (size_t)kc * a.vr_rs);
                    pp[u] = key < a.nkv_rows ? pl[kc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc[0] = fmaf(pp[u], bflo(vv[u].x), acc[0]); acc[1] = fmaf(pp[u], bfhi(vv[u].x), acc[1]);
                    acc[2] = fmaf(pp[u], bflo(vv[u].y), acc[2]); acc[3] = fmaf(pp[u], bfhi(vv[u].y), acc[3]);
                    acc[4] = fmaf(pp[u], bflo(vv[u].z), acc[4]); acc[5] = fmaf(pp[u], bfhi(vv[u].z), acc[5]);
                    acc[6] = fmaf(pp[u], bflo(vv[u].w), acc[6]); acc[7] = fmaf(pp[u], bfhi(vv[u].w), acc[7]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] += __shfl_xor(acc[e], 8, 64);
                acc[e] += __shfl_xor(acc[e], 16, 64);
                acc[e] += __shfl_xor(acc[e], 32, 64);
            }
            if (kg == 0) {
                const float inv = 1.f / ls;
                uint4 o;
                o.x = pack2bf(acc[0] * inv, acc[1] * inv); o.y = pack2bf(acc[2] * inv, acc[3] * inv);
                o.z = pack2bf(acc[4] * inv, acc[5] * inv); o.w = pack2bf(acc[6] * inv, acc[7] * inv);
                *reinterpret_cast<uint4*>(a.O + (size_t)b * a.o_bs + (size_t)kvh * a.o_hs + (size_t)row * a.o_rs + c8 * 8) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next row overwrites pl)
        }
        return;
    }

    // ---------------- MFMA workgroup: query rows [qb * 192, + 192) of one (frame, head) ----------------
    const int bh = quad * 4 + rq / nqb, qb = rq - (rq / nqb) * nqb;
    if (bh >= BH) return;                                        // (workgroup-uniform)
    const int b = bh / a.kv_heads, kvh = bh - b * a.kv_heads;
    const int q0 = qb * BIG_ROWS + wid * BIG_QW;
    const int rem = a.nkv_rows % BKV, nkv_main = a.nkv_rows - rem, nt = nkv_main / BKV;
    const bf16_t* kbase = a.K + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
    const bf16_t* vbase = a.Vrow + (size_t)b * a.vr_bs + (size_t)kvh * a.vr_hs;

    // ---- LDS-DMA: per tile 8 K pieces (8 rows x 128 B, XOR-swizzled on the source) + 8 V pieces (16 keys x 64 B of one d-half, halves swapped
    //      for odd key quads); wave w issues pieces 2 w, 2 w + 1 of each
    // (sources as 32-bit byte offsets from the wave-uniform head bases: scalar base + one offset register per piece — as 64-bit pointers the four of
    // them were what the 168-register build spilled and reloaded in every tile)
    uint32_t ksrc[2], vsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pi = 2 * wid + j;
        const int kr = pi * 8 + (lane >> 3);
        ksrc[j] = ((uint32_t)kr * (uint32_t)a.k_rs + (uint32_t)(((lane & 7) ^ ((kr >> 1) & 7)) << 3)) * 2u;
        const int vkey = (pi & 3) * 16 + (lane >> 2);
        vsrc[j] = ((uint32_t)vkey * (uint32_t)a.vr_rs + (uint32_t)((pi >> 2) * 32 + (((lane & 3) ^ (((vkey >> 2) & 1) << 1)) << 3))) * 2u;
    }
    // (round 6: through buffer descriptors — scalar head base + the per-lane 32-bit piece offset + a SCALAR tile offset: the global_load_lds form kept the four
    // offsets as zero-extended 64-bit pairs, and the registers that cost were reloaded from scratch inside the tile loop behind an s_waitcnt vmcnt(0), i.e. every
    // tile waited for the K / V pieces it had just requested)
    const uint32_t kstep = (uint32_t)BKV * (uint32_t)a.k_rs * 2u, vstep = (uint32_t)BKV * (uint32_t)a.vr_rs * 2u;      // bytes per key tile
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0, (int)((uint32_t)a.nkv_rows * (uint32_t)a.k_rs * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0, (int)((uint32_t)a.nkv_rows * (uint32_t)a.vr_rs * 2u), 0x00020000);
    auto issue = [&](int t, int stage) {
        char* st = smem + stage * BSTAGE + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(st + j * 1024), 16, ksrc[j], t * kstep, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(st + 8192 + j * 1024), 16, vsrc[j], t * vstep, 0, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nt) issue(t, t);

    // ---- the first trailing key's K row and V row are requested BEFORE the Q fragments (round 6): behind them they cost every workgroup a second, fully exposed
    //      memory round trip in front of its first tile (knock-out of the trailing key: 366 -> 344 us, profiles/r06_attn_big_knockouts.txt)
    const bool skip_rem = a.dbg == 3 || a.dbg == 5;
    uint4 rk0[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    uint2 rv0[4] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
    if (rem > 0 && !skip_rem) {
        const bf16_t* kp = kbase + (size_t)nkv_main * a.k_rs + g * 8;
        const bf16_t* vp = vbase + (size_t)nkv_main * a.vr_rs + 4 * g;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) rk0[ks] = *reinterpret_cast<const uint4*>(kp + ks * 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) rv0[dt] = *reinterpret_cast<const uint2*>(vp + dt * 16);
    }
    // ---- Q fragments, scaled once: q' = bf16(q * scale * log2 e)
    bf16x8_t qf[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bf16_t* qp = a.Q + (size_t)b * a.q_bs + (size_t)kvh * a.q_hs + (size_t)(q0 + i * 16 + r16) * a.q_rs + g * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 q4 = a.dbg == 5 ? make_uint4(0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u) : *reinterpret_cast<const uint4*>(qp + ks * 32);
            union { bf16x8_t v; uint32_t u[4]; } o;
            o.u[0] = pack2bf(bflo(q4.x) * sc, bfhi(q4.x) * sc);
            o.u[1] = pack2bf(bflo(q4.y) * sc, bfhi(q4.y) * sc);
            o.u[2] = pack2bf(bflo(q4.z) * sc, bfhi(q4.z) * sc);
            o.u[3] = pack2bf(bflo(q4.w) * sc, bfhi(q4.w) * sc);
            qf[i][ks] = o.v;
        }
    }
    f32x4_t oacc[4][3], lacc[3], cinit[3];
    float m[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        m[i] = 0.f;
        lacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    bool first = true;

    // ---- trailing keys (nkv % 64 of them) on the VALU: they initialise the softmax state
    // (a.dbg 3 / 4 / 5, timing only — what a persistent form could hide at most: 3 = without these trailing keys, 4 = without the output stores, 5 = both and
    // the Q fragments not fetched)
    // The FIRST trailing key opens the softmax state, and for it everything but the three dot products is known: reference m = its own score, p = 2^0 = 1, so
    // l = 1 and O = its V row (round 6; the general update below spent ~250 VALU instructions per wave on it — one key for the price of a 64-key tile's VALU work,
    // in a kernel whose VALU pipe, not its matrix pipe, is the busy one: knock-out of the trailing key 336 -> 307 us).  The dots run on v_dot2.
    int jt0 = 0;
    if (rem > 0 && !skip_rem) {
        float vf[4][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            vf[dt][0] = bflo(rv0[dt].x); vf[dt][1] = bfhi(rv0[dt].x); vf[dt][2] = bflo(rv0[dt].y); vf[dt][3] = bfhi(rv0[dt].y);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float dot = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { bf16x8_t v; uint32_t u[4]; } qq;
                qq.v = qf[i][ks];
                dot = dot2_el(qq.u[0], rk0[ks].x, dot); dot = dot2_el(qq.u[1], rk0[ks].y, dot);
                dot = dot2_el(qq.u[2], rk0[ks].z, dot); dot = dot2_el(qq.u[3], rk0[ks].w, dot);
            }
            dot += __shfl_xor(dot, 16, 64);                      // the four lanes (q, g = 0..3) hold 16 d each
            dot += __shfl_xor(dot, 32, 64);
            m[i] = dot;
            lacc[i] = f32x4_t{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt][i] = f32x4_t{vf[dt][0], vf[dt][1], vf[dt][2], vf[dt][3]};
        }
        first = false;
        jt0 = 1;
    }
    for (int jt = jt0; jt < (skip_rem ? 0 : rem); ++jt) {
        const int kv = nkv_main + jt;
        const bf16_t* kp = kbase + (size_t)kv * a.k_rs + g * 8;
        uint32_t ku[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 kk = *reinterpret_cast<const uint4*>(kp + ks * 32);
            ku[ks][0] = kk.x; ku[ks][1] = kk.y; ku[ks][2] = kk.z; ku[ks][3] = kk.w;
        }
        const bf16_t* vp = vbase + (size_t)kv * a.vr_rs + 4 * g;
        uint2 vv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vv[dt] = *reinterpret_cast<const uint2*>(vp + dt * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float dot = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { bf16x8_t v; uint32_t u[4]; } qq;
                qq.v = qf[i][ks];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dot = fmaf(bflo(qq.u[e]), bflo(ku[ks][e]), dot);
                    dot = fmaf(bfhi(qq.u[e]), bfhi(ku[ks][e]), dot);
                }
            }
            dot += __shfl_xor(dot, 16, 64);                      // the four lanes (q, g = 0..3) hold 16 d each
            dot += __shfl_xor(dot, 32, 64);
            const float mnew = first ? dot : fmaxf(m[i], dot);
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m[i] - mnew);
            const float p = __builtin_amdgcn_exp2f(dot - mnew);
            m[i] = mnew;
#pragma unroll
            for (int e = 0; e < 4; ++e) lacc[i][e] = fmaf(lacc[i][e], alpha, p);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                oacc[dt][i][0] = fmaf(p, bflo(vv[dt].x), oacc[dt][i][0] * alpha);
                oacc[dt][i][1] = fmaf(p, bfhi(vv[dt].x), oacc[dt][i][1] * alpha);
                oacc[dt][i][2] = fmaf(p, bflo(vv[dt].y), oacc[dt][i][2] * alpha);
                oacc[dt][i][3] = fmaf(p, bfhi(vv[dt].y), oacc[dt][i][3] * alpha);
            }
        }
        first = false;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float ci = first ? 0.f : -m[i];
        cinit[i] = f32x4_t{ci, ci, ci, ci};
    }
    union { bf16x8_t v; uint32_t u[4]; } ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = TRACE_EL_ONE2;

    // lane offsets: K fragment rows r16 of a 16-key block (k-step 0; k-step 1 is ^ 64), V transpose reads of d-tiles with (dt & 1) = 0 / 1
    const int koff = kswz<HD>(r16, g);                          // (rows j * 16 + r16: (row >> 1) & 7 does not depend on j)
    const int vrow = (4 * g + (r16 >> 2)) * 64 + (r16 & 3) * 8;
    const int vtr0 = vrow + (g & 1) * 32, vtr1 = vrow + ((g & 1) ^ 1) * 32;
    // LDS byte address of stage 0's V image (the asm reads take the 32-bit LDS address, not a generic pointer)
    const uint32_t vlds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + 8192u;

    int stage = 0, stage_n = NST - 1;
    for (int t = 0; t < nt; ++t) {
        // tile t landed: this wave's pieces by its own counted vmcnt (4 pieces per tile; the younger tiles may stay in flight), everybody's by the
        // barrier; the barrier also says every wave is done reading tile t - 1, whose stage the DMA issued next overwrites
        const int ahead = min(NST - 2, nt - 1 - t);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt && a.dbg != 1) issue(t + NST - 1, stage_n);
        if (a.dbg != 2) {
            const char* kb = smem + stage * BSTAGE;
            // the tile in two halves of 32 keys (kv-tiles 2 kp, 2 kp + 1): only 24 score registers are live at a time — with the whole tile's 48 the kernel
            // needs 192 registers = two waves per SIMD, and its first hardware run (profiles/r05_attn_big_first_run.txt, r05_pmc_attn.txt) was 23 % SLOWER
            // than the 4 x 32-row kernel although it issues 13 % fewer instructions: a workgroup lives for 9 key tiles, its prologue / epilogue / dispatch
            // bubbles only hide under other resident workgroups, and two per CU (issue slots 43 % busy against 73 %) are not enough
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                // ---- S^T = K . Q^T - m for 32 keys
                f32x4_t S[2][3];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * kp + jj;
                    const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t*>(kb + j * 2048 + koff);
                    const bf16x8_t kf1 = *reinterpret_cast<const bf16x8_t*>(kb + j * 2048 + (koff ^ 64));
#pragma unroll
                    for (int i = 0; i < 3; ++i) mfma16_from(S[jj][i], kf0, qf[i][0], cinit[i]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) S[jj][i] = mfma16(kf1, qf[i][1], S[jj][i]);
                }
                // the V fragments of these keys: their LDS latency runs under the checks and exponentials below
                u32x2_t vq[4][2];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    tr_read2(vq[dt][0], vq[dt][1], vlds + stage * BSTAGE + (dt >> 1) * 4096 + (2 * kp) * 1024 + ((dt & 1) ? vtr1 : vtr0));
                // ---- does any row need its reference moved?  (integer compare of the bit patterns; the exact row maxima are taken inside the slow path)
                const int over = imax3(bits_max8(S[0][0], S[1][0]), bits_max8(S[0][1], S[1][1]), bits_max8(S[0][2], S[1][2]));
                if (__any(first || over > __float_as_int(BIG_THR))) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float mloc = fmaxf(fmaxf(fmaxf(S[0][i][0], S[0][i][1]), fmaxf(S[0][i][2], S[0][i][3])),
                                                 fmaxf(fmaxf(S[1][i][0], S[1][i][1]), fmaxf(S[1][i][2], S[1][i][3])));
                        float mx = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                        const bool need = first || mx > BIG_THR;     // the same in the four lanes of a query row
                        const float d = need ? mx : 0.f;
                        const float f = first ? 0.f : __builtin_amdgcn_exp2f(-d);       // (first keys: O = l = 0, and 2^-d may overflow)
                        m[i] += d;
#pragma unroll
                        for (int e = 0; e < 4; ++e) lacc[i][e] *= f;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) oacc[dt][i][e] *= f;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int e = 0; e < 4; ++e) S[jj][i][e] -= d;
                        const float ci = -m[i];
                        cinit[i] = f32x4_t{ci, ci, ci, ci};
                    }
                    first = false;
                }
                // ---- p = 2^S, O^T += V^T . P, l += 1 . P
                union { bf16x8_t v; uint32_t u[4]; } pf[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4_t& s0 = S[0][i];
                    const f32x4_t& s1 = S[1][i];
                    pf[i].u[0] = pack2bf(__builtin_amdgcn_exp2f(s0[0]), __builtin_amdgcn_exp2f(s0[1]));
                    pf[i].u[1] = pack2bf(__builtin_amdgcn_exp2f(s0[2]), __builtin_amdgcn_exp2f(s0[3]));
                    pf[i].u[2] = pack2bf(__builtin_amdgcn_exp2f(s1[0]), __builtin_amdgcn_exp2f(s1[1]));
                    pf[i].u[3] = pack2bf(__builtin_amdgcn_exp2f(s1[2]), __builtin_amdgcn_exp2f(s1[3]));
                }
                tr_wait(vq);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    union { bf16x8_t v; u32x2_t q[2]; } u;
                    u.q[0] = vq[dt][0]; u.q[1] = vq[dt][1];
#pragma unroll
                    for (int i = 0; i < 3; ++i) oacc[dt][i] = mfma16(u.v, pf[i].v, oacc[dt][i]);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) lacc[i] = mfma16(ones.v, pf[i].v, lacc[i]);
            }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
        stage_n = stage_n + 1 == NST ? 0 : stage_n + 1;
    }

    if ((a.dbg == 4 || a.dbg == 5) && lacc[0][0] != 12345.678f) return;          // (the condition keeps the accumulators live)
    // Output rows, 16 bytes per store (round 6; the same bytes as the 8-byte form): lane (q, g) holds d = dt 16 + 4 g .. + 3 of query row q for every d-tile dt.  One
    // v_permlane16_swap per register of a d-tile pair (dt, dt + 1) hands the dt piece of the odd-g lane to its even-g neighbour and the dt + 1 piece of the even-g lane
    // to the odd one: even g then holds d = dt 16 + 4 g .. + 7, odd g holds d = (dt + 1) 16 + 4 (g - 1) .. + 7 — 6 stores per lane instead of 12.  The store tail is
    // issue-bound, not bandwidth-bound (knock-out of the stores: 366 -> 334 us).
    const bool wide = (a.o_rs % 8 == 0) && (a.o_hs % 8 == 0) && (a.o_bs % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.O) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.f / lacc[i][0];
        bf16_t* orow = a.O + (size_t)b * a.o_bs + (size_t)kvh * a.o_hs + (size_t)(q0 + i * 16 + r16) * a.o_rs;
        uint2 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt].x = pack2bf(oacc[dt][i][0] * inv, oacc[dt][i][1] * inv);
            o[dt].y = pack2bf(oacc[dt][i][2] * inv, oacc[dt][i][3] * inv);
        }
        if (wide) {
#pragma unroll
            for (int dp = 0; dp < 4; dp += 2) {
                swap16_rows(o[dp].x, o[dp + 1].x);
                swap16_rows(o[dp].y, o[dp + 1].y);
                const int d0 = (g & 1) ? (dp + 1) * 16 + 4 * (g - 1) : dp * 16 + 4 * g;
                *reinterpret_cast<uint4*>(orow + d0) = make_uint4(o[dp].x, o[dp].y, o[dp + 1].x, o[dp + 1].y);
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<uint2*>(orow + 4 * g + dt * 16) = o[dt];
        }
    }
}

// V [rows, HD] (row stride src_rs) -> V^T [HD, rows_pad] (row stride dst_rs), zero-filled for rows >= n.
// One block per 64-row tile: 16-byte loads -> LDS -> 16-byte stores along the token axis.
// perm = 1: within every 16 keys the destination order is [0-3, 8-11, 4-7, 12-15] (attn_vit_dma_kernel's PV operand layout).
template <int HD>
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ src, long src_bs, long src_hs, int src_rs,
                                                          bf16_t* __restrict__ dst, long dst_bs, long dst_hs, int dst_rs,
                                                          int n, int perm) {
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
            // source rows of destination positions 2e, 2e+1 of this 8-key chunk
            const int r0 = perm ? 16 * (tc >> 1) + 4 * (tc & 1) + ((2 * e) & 3) + 8 * ((2 * e) >> 2) : tc * 8 + 2 * e;
            const uint32_t lo = tile[r0 * LROW + dd], hi = tile[(r0 + 1) * LROW + dd];
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
    static LdsGrant grant;
    if (!grant_dynamic_lds(grant, reinterpret_cast<const void*>(attn_kernel<HD, GQA>), (int)lds)) return TRACE_ERR_HIP;
    const int nqt = (a.nq_rows + 31) / 32;
    const long nblk = (long)(GQA ? nqt : (nqt + 3) / 4) * a.kv_heads * a.batch;
    if (nblk > 0x7fffffffL) return TRACE_ERR_ARG;
    hipLaunchKernelGGL((attn_kernel<HD, GQA>), dim3((unsigned)nblk), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
}  // namespace

int g_attn_pf_debug = 0;   // microbenchmark-only knock-outs of the prefill-shaped kernels: 1 = no K/V loads after tile 0, 2 = no tile math

// The LDS-DMA kernel takes V^T in transpose_v's permuted layout (v_perm = 1) or V row-major (v_perm = 2, attn_vit_rowmajor_v()) and needs
// whole key tiles plus at most TAILV trailing keys
bool attn_vit_rowmajor_v() { return g_attn_pf_debug != 6; }    // trace_op_set_gemm_variant(116): the transposed-V path, for A/B runs
bool attn_vit_wants_perm(int nkv_rows, bool has_vrow) {
    const int rem = nkv_rows % BKV;
    return (g_attn_pf_debug < 5 || g_attn_pf_debug == 6) && nkv_rows >= 2 * BKV && (rem == 0 || (has_vrow && rem <= TAILV));
}

int g_attn_vit_big = 2;    // the round-5 192-row kernel for shapes it takes (trace_op_set_gemm_variant(190 + x): 0 = the 4 x 32-row kernel, 1 = 4-stage ring (two workgroups
                           // per CU), 2 = 3-stage ring (three per CU: the default))
// v_perm == 2 (row-major V), >= 192 query rows of which 192 leaves at most BIG_MAXLEFT over, whole key tiles + a short tail, keys fit the leftover path's LDS rows
static bool attn_vit_big_ok(const AttnArgs& a) {
    return g_attn_vit_big && a.v_perm == 2 && !a.causal && a.nq_rows >= BIG_ROWS && a.nq_rows % BIG_ROWS <= BIG_MAXLEFT && a.nkv_rows >= BKV &&
           a.nkv_rows % BKV <= TAILV && a.nkv_rows <= BIG_MAXKV && !(a.q_rs % 8) && !(a.q_hs % 8) && !(a.q_bs % 8) && !(a.k_rs % 8) && !(a.k_hs % 8) &&
           !(a.k_bs % 8) && !(a.o_rs % 8) && !(a.o_hs % 8) && !(a.o_bs % 8);
}
int launch_attn_vit(const AttnArgs& a_, hipStream_t s) {
    AttnArgs a = a_;
    a.dbg = g_attn_pf_debug;
    if (a.heads != a.kv_heads || a.nq_rows <= 0 || a.nkv_rows <= 0 || (a.v_rs % 64)) return TRACE_ERR_ARG;
    if (a.v_perm) {
        if (a.causal || !attn_vit_wants_perm(a.nkv_rows, a.Vrow != nullptr)) return TRACE_ERR_ARG;
        if (a.v_perm == 2 && (!a.Vrow || (a.vr_rs % 8) || (a.vr_hs % 8) || (a.vr_bs % 8))) return TRACE_ERR_ARG;
        if (attn_vit_big_ok(a)) {
            const int nqb = a.nq_rows / BIG_ROWS, per = 4 * nqb + (a.nq_rows % BIG_ROWS ? 1 : 0);
            const long nblk = (long)(((long)a.batch * a.kv_heads + 3) / 4) * per;
            if (nblk > 0x7fffffffL) return TRACE_ERR_ARG;
            static LdsGrant grant3, grant4;
            if (g_attn_vit_big == 2) {
                if (!grant_dynamic_lds(grant3, reinterpret_cast<const void*>(attn_vit_big_kernel<3>), 3 * BSTAGE)) return TRACE_ERR_HIP;
                hipLaunchKernelGGL(attn_vit_big_kernel<3>, dim3((unsigned)nblk), dim3(256), 3 * BSTAGE, s, a);
            } else {
                if (!grant_dynamic_lds(grant4, reinterpret_cast<const void*>(attn_vit_big_kernel<4>), 4 * BSTAGE)) return TRACE_ERR_HIP;
                hipLaunchKernelGGL(attn_vit_big_kernel<4>, dim3((unsigned)nblk), dim3(256), 4 * BSTAGE, s, a);
            }
            return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
        }
        static LdsGrant grant_t, grant_r;
        if (!grant_dynamic_lds(grant_t, reinterpret_cast<const void*>(attn_vit_dma_kernel<false>), DSTAGES * DSTAGE) ||
            !grant_dynamic_lds(grant_r, reinterpret_cast<const void*>(attn_vit_dma_kernel<true>), DSTAGES * DSTAGE)) return TRACE_ERR_HIP;
        const int nqt = (a.nq_rows + 31) / 32;
        const long nblk = (long)((nqt + 3) / 4) * a.kv_heads * a.batch;
        if (nblk > 0x7fffffffL) return TRACE_ERR_ARG;
        if (a.v_perm == 2) hipLaunchKernelGGL(attn_vit_dma_kernel<true>, dim3((unsigned)nblk), dim3(256), DSTAGES * DSTAGE, s, a);
        else hipLaunchKernelGGL(attn_vit_dma_kernel<false>, dim3((unsigned)nblk), dim3(256), DSTAGES * DSTAGE, s, a);
        return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
    }
    return launch_attn<64, false>(a, s);
}

int launch_attn_prefill(const AttnArgs& a, hipStream_t s) {
    if (a.heads != 4 * a.kv_heads || a.nq_rows <= 0 || a.nkv_rows < a.nq_rows || (a.v_rs % 64)) return TRACE_ERR_ARG;
    return launch_attn<128, true>(a, s);
}

int launch_transpose_v(const bf16_t* src, long src_bs, long src_hs, int src_rs, bf16_t* dst, long dst_bs, long dst_hs,
                       int dst_rs, int n, int hd, int heads, int batch, hipStream_t s, int perm) {
    if (dst_rs % 64 || dst_rs < n || (hd != 64 && hd != 128)) return TRACE_ERR_ARG;
    dim3 grid((n + 63) / 64, heads, batch);
    if (hd == 64) hipLaunchKernelGGL(transpose_v_kernel<64>, grid, dim3(256), 0, s, src, src_bs, src_hs, src_rs, dst, dst_bs, dst_hs, dst_rs, n, perm);
    else hipLaunchKernelGGL(transpose_v_kernel<128>, grid, dim3(256), 0, s, src, src_bs, src_hs, src_rs, dst, dst_bs, dst_hs, dst_rs, n, perm);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
