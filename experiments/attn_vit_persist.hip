// NOT BUILT.  Persistent, LDS-resident form of the ViT attention (round 2 experiment; see experiments/README.md).
// It drops into trace_amd/csrc/attn.hip after attn_vit_dma_kernel (it uses that file's kswz / glds16 / DSTAGE / RESCALE_THR) and was
// launched as  attn_vit_persist_kernel<<<min(frames*heads, #CUs), 768, 9*16384>>>(args).  Correct (the GPU parity tests passed with it
// in place) but slower than the 4-wave LDS-DMA ring kernel: 557 us vs 407 us per 170-frame launch.
// =====================================================================================================================
// ViT attention, persistent form (the one the engine runs at the CLIP-ViT-L/14-336 geometry).
// What the probes of the two kernels above showed (tools/attn_vit_probe.py, tools/attn_vit_pmc.py; 170 frames):
//   * each 4-wave block streams its head's whole K/V (148 KB) for only 128 query rows, five blocks per head: 2 GB per launch
//     through the CUs' memory path; the stream alone (no tile math) takes 253 us whether it is staged through registers or by
//     LDS-DMA — with two tiles (32 KB) in flight per block and ~3 us to an L2 miss, a CU cannot have more than ~100 KB under way;
//   * on a SIMD, VALU issue and MFMA execution add up rather than overlap (VALU port 62 % + MFMA 33 % busy), so what is
//     left is to cut instructions per tile (done above: 490 -> ~120) and to stop paying the K/V stream five times.
// Here ONE 12-wave workgroup per CU keeps the K/V of a whole head resident in LDS (9 tiles x 16 KB = 144 KB) and runs all 19
// query tiles against it in two passes (12 + 7 tiles; with the cyclic wave->SIMD placement that is 5/5/5/4 tile-passes per SIMD),
// so every K/V byte is fetched once (0.4 GB per launch).  The workgroup is persistent and walks heads: during the second pass,
// whose query tiles occupy waves 0-6 only, waves 8-11 — idle in that pass — prepare their next query tile and issue the NEXT head's
// tiles by LDS-DMA into each stage as soon as the barrier says every wave has finished reading it, so the next head's first pass
// starts on data that has been in flight for a whole pass.  Only those four waves ever issue DMA or wait on vmcnt (exact counts:
// they drain their own output stores first); the other eight see K/V only through the one barrier per tile step.
constexpr int PWAVES = 12, PLOAD0 = 8, PSTAGES = 9;

__device__ __forceinline__ void vit_wait_vm(int n_tiles_after) {          // 4 DMA pieces per tile per loader wave
    switch (n_tiles_after) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    }
}

// one key tile of attn_vit_dma_kernel's loop (see there): S = K.q' - m straight out of the MFMA, lazy reference, exp2, PV
__device__ __forceinline__ void vit_tile(const char* kb, const char* vb, const bf16x8_t (&qf)[4], f32x16_t (&oacc)[2], f32x16_t& cinit,
                                         float& m, float& l, bool& first, int c32, int h) {
    constexpr int HD = 64;
    f32x16_t S[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int row = st * 32 + c32;
#pragma unroll
        for (int s_ = 0; s_ < HD / 16; ++s_) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + kswz<HD>(row, s_ * 2 + h));
            if (s_ == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(S[st]) : "v"(kf), "v"(qf[0]), "v"(cinit));
            else S[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s_], S[st], 0, 0, 0);
        }
    }
    float mt = fmaxf(S[0][0], S[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, S[0][r]), S[1][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const bool need = first || mt > RESCALE_THR;
    if (__any(need)) {
        const float d = need ? mt : 0.f;
        const float f = first ? 0.f : __builtin_amdgcn_exp2f(-d);
        m += d;
        l *= f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
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
        for (int ht = 0; ht < 2; ++ht) {
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vb + kswz<HD>(ht * 32 + c32, ks * 2 + h));
            oacc[ht] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, oacc[ht], 0, 0, 0);
        }
    }
}

// query tile `qt` of head item `item`: scaled Q fragments, state initialised from the trailing keys (VALU)
__device__ __forceinline__ bool vit_prep(const AttnArgs& a, int item, int qt, int nkv_main, int rem, float sc, int c32, int h,
                                         bf16x8_t (&qf)[4], f32x16_t (&oacc)[2], f32x16_t& cinit, float& m, float& l, bool& first) {
    constexpr int HD = 64;
    const int q0 = qt * 32;
    if (q0 >= a.nq_rows) return false;
    const int b = item / a.kv_heads, kvh = item - b * a.kv_heads;
    const int qr = min(q0 + c32, a.nq_rows - 1);
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    m = 0.f; l = 0.f; first = true;
    const bf16_t* kbase = a.K + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
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
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const uint2 vv = *reinterpret_cast<const uint2*>(vp + ht * 32 + 8 * rg);
                oacc[ht][rg * 4 + 0] = fmaf(p, bflo(vv.x), oacc[ht][rg * 4 + 0] * alpha);
                oacc[ht][rg * 4 + 1] = fmaf(p, bfhi(vv.x), oacc[ht][rg * 4 + 1] * alpha);
                oacc[ht][rg * 4 + 2] = fmaf(p, bflo(vv.y), oacc[ht][rg * 4 + 2] * alpha);
                oacc[ht][rg * 4 + 3] = fmaf(p, bfhi(vv.y), oacc[ht][rg * 4 + 3] * alpha);
            }
    }
    const float ci = first ? 0.f : -m;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = ci;
    return true;
}

__device__ __forceinline__ void vit_store(const AttnArgs& a, int item, int qt, const f32x16_t (&oacc)[2], float l, int c32, int h) {
    const int qabs = qt * 32 + c32;
    const float lt = l + __shfl_xor(l, 32, 64);
    if (qabs >= a.nq_rows) return;
    const float inv = 1.f / lt;
    const int b = item / a.kv_heads, kvh = item - b * a.kv_heads;
    bf16_t* op = a.O + (size_t)b * a.o_bs + (size_t)kvh * a.o_hs + (size_t)qabs * a.o_rs;
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            uint2 o;
            o.x = pack2bf(oacc[ht][rg * 4 + 0] * inv, oacc[ht][rg * 4 + 1] * inv);
            o.y = pack2bf(oacc[ht][rg * 4 + 2] * inv, oacc[ht][rg * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(op + ht * 32 + 8 * rg + 4 * h) = o;
        }
}

__global__ __launch_bounds__(PWAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_vit_persist_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    const int rem = a.nkv_rows % BKV, nkv_main = a.nkv_rows - rem, nt = nkv_main / BKV;      // nt <= PSTAGES
    const int nitems = a.kv_heads * a.batch;
    const bool loader = wid >= PLOAD0;
    const float sc = a.scale * 1.4426950408889634f;
    // DMA lane offsets of loader wave lw: pieces 2 lw, 2 lw + 1 of the 8 K and the 8 V^T pieces of a tile (8 rows x 128 B each)
    int koff[2], voff[2];
    const int lw = loader ? wid - PLOAD0 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (2 * lw + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        koff[j] = r * a.k_rs + c * 8;
        voff[j] = r * a.v_rs + c * 8;
    }
#define VIT_ISSUE(ITEM_, T_)                                                                                   \
    {                                                                                                          \
        const int b_ = (ITEM_) / a.kv_heads, kvh_ = (ITEM_) - b_ * a.kv_heads;                                 \
        const bf16_t* kb_ = a.K + (size_t)b_ * a.k_bs + (size_t)kvh_ * a.k_hs + (size_t)(T_) * BKV * a.k_rs;  \
        const bf16_t* vb_ = a.V + (size_t)b_ * a.v_bs + (size_t)kvh_ * a.v_hs + (T_) * BKV;                    \
        char* st_ = smem + (T_) * DSTAGE + (2 * lw) * 1024;                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
            glds16(kb_ + koff[j], st_ + j * 1024);                                                             \
            glds16(vb_ + voff[j], st_ + 8192 + j * 1024);                                                      \
        }                                                                                                      \
    }

    bf16x8_t qf[4];
    f32x16_t oacc[2], cinit;
    float m = 0.f, l = 0.f;
    bool first = true, have = false;         // have: the registers hold a prepared query tile
    // one phase = (head item, pass); every wave prepares at ONE place (phase start): a compute wave the query tile it runs in this
    // phase, a loader wave — idle in second passes — its first-pass tile of the NEXT head (and, once, of the first head)
    const int nphases = 2 * ((nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
#pragma unroll 1
    for (int ph = 0; ph < nphases; ++ph) {
        const int pass = ph & 1;
        const int item = blockIdx.x + (ph >> 1) * gridDim.x, next = item + gridDim.x;
        const bool has_next = next < nitems;
        {
            const int tgt = loader ? (pass == 1 ? (has_next ? next : -1) : (ph == 0 ? item : -1)) : item;
            const int qt = loader ? wid : pass * PWAVES + wid;
            // loader, start of a second pass (or of the kernel): drain my output stores — the counted waits of the first pass must see DMA
            // pieces only.  (Not at the start of later first passes: the next head's tiles are in flight then, that is the point.)
            const bool drain = loader && (pass == 1 || ph == 0);
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tgt >= 0) have = vit_prep(a, tgt, qt, nkv_main, rem, sc, c32, h, qf, oacc, cinit, m, l, first);
            else if (!(loader && pass == 0)) have = false;
            if (drain) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the prepared tile's own loads
                if (ph == 0 && a.dbg != 1)
                    for (int t = 0; t < nt; ++t) VIT_ISSUE(item, t)
            }
        }
        const bool mine = have && (pass == 0 || !loader);
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            if (pass == 0 && loader) vit_wait_vm(nt - 1 - t);          // my pieces of tile t are in LDS (later tiles may be in flight)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // second pass: every wave is past tile t-1 -> its stage takes the next head's tile t-1
            if (pass == 1 && loader && has_next && t >= 1 && a.dbg != 1) VIT_ISSUE(next, t - 1)
            if (mine && a.dbg != 2) vit_tile(smem + t * DSTAGE, smem + t * DSTAGE + 8192, qf, oacc, cinit, m, l, first, c32, h);
        }
        if (pass == 1) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (loader && has_next && a.dbg != 1) VIT_ISSUE(next, nt - 1)
        }
        if (mine) vit_store(a, item, pass * PWAVES + wid, oacc, l, c32, h);
    }
#undef VIT_ISSUE
}

