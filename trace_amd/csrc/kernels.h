// Internal launcher interface between the kernel translation units and the engine (engine.hip).
// Everything here takes raw device pointers + an explicit stream; the public C ABI is include/trace_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "common.h"

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: one process may hold contexts on several GPUs, so the grant is
// tracked per device (a bit per device id), not once per process; safe from both launch threads of the pipeline (a double grant is idempotent).
struct LdsGrant { std::atomic<uint32_t> devs{0}; };
inline bool grant_dynamic_lds(LdsGrant& g, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return false;
    const uint32_t bit = 1u << dev;
    if (g.devs.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    g.devs.fetch_or(bit, std::memory_order_release);
    return true;
}

// the same for kernels whose dynamic-LDS request varies per launch: the largest size granted so far, per device
struct LdsGrantSized { std::atomic<size_t> bytes[32]; LdsGrantSized() { for (auto& b : bytes) b.store(0); } };
inline bool grant_dynamic_lds(LdsGrantSized& g, const void* fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return false;
    if (bytes <= g.bytes[dev].load(std::memory_order_acquire)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    size_t cur = g.bytes[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !g.bytes[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
    return true;
}

constexpr int SK_ROWS = 256;     // largest decode batch = row stride of the fp32 k-chunk partial buffers [ks][SK_ROWS][N]
constexpr int SKINNY_ROWS = 64;  // rows one skinny (activations-parked-in-LDS) GEMV takes; larger batches run the split-K MFMA GEMM (gemm.hip)
enum { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_QUICKGELU = 2, EPI_SWIGLU = 3, EPI_PARTIAL = 4 };   // PARTIAL: decode GEMV only

struct GemmArgs {
    const bf16_t* A; int lda;       // [M,K] activations, row stride lda (elements)
    const bf16_t* W; int ldw;       // [N,K] weights (nn.Linear layout)
    bf16_t* C; int ldc;             // [M,N] (or [M,N/2] for EPI_SWIGLU)
    const bf16_t* bias;             // [N] or null
    const bf16_t* R; int ldr;       // residual [M,N] for EPI_RESIDUAL (may alias C)
    int M, N, K;
    unsigned long long* trace;      // profiling only (tools/gemm_trace.py): 8 x u64 per workgroup, or null
    // fp8 = 1: A and W point at e4m3 BYTES (lda / ldw in bytes, K % 128 == 0); C = (A8 . W8^T) * sa[m] * sw[n] (+ epilogue)
    int fp8; const float* sa; const float* sw;
    int opt;                        // A/B switches of a kernel (0 = shipped behaviour)
    // EPI_PARTIAL (decode batches above SKINNY_ROWS, 128x128 tiles only): K is cut in `ks` equal chunks of whole K-tiles, one workgroup per
    // (tile, chunk); chunk c leaves its fp32 accumulators in part[c][m][n] (row stride SK_ROWS x N) — the partial rows the decode consumers
    // (qkv_finish, add_rmsnorm) sum on load.  C / bias / R unused.
    float* part; int ks;
    // w_tiled (128x128 tiles, bf16 only; the decode GEMMs): W is the decode copy made by launch_tile_pack ([N/16][K/64][64 lanes][16]) — a
    // workgroup's weight stream is then 8 contiguous runs of 2 KB blocks instead of 128 row pieces of 128 B at an 8 KB stride (DRAM page
    // locality, as for the GEMV).  Bit 2: a 4-stage K-tile ring (three tiles in flight) instead of the double buffer.  (Bit 1 was a non-temporal
    // hint on the weight DMA: 11.58 vs 11.15 ms per 128-sequence step, profiles/r03_decode_gemm_ab_b128.txt — removed.)
    int w_tiled;
};
int launch_gemm_bf16(const GemmArgs& p, int epi, hipStream_t s);
int gemm_partial_ks(int N, int K);                                // K-chunks launch_gemm_bf16(EPI_PARTIAL) should be given for an [<= 128, K] x [N, K]^T product
int launch_gemm_ldr(const GemmArgs& p, int epi, hipStream_t s);    // gemm_ldr.hip: 256x256 tiles, 8 MFMA + 4 loader waves (N % 256 == 0)
int launch_gemm_pers(const GemmArgs& p, int epi, hipStream_t s);   // gemm_pers.hip: the same tile, persistent workgroups, register epilogue (bf16, K >= 128)
int launch_gemm_w4(const GemmArgs& p, int epi, hipStream_t s);     // gemm_w4.hip: the persistent 256x256 tile on 4 waves of 128x128 (bf16, K >= 192)
int gemm_pers_plan(hipStream_t s, int total, int** ctr, int* nblk);   // the stream's ticket counters + persistent grid size (shared by gemm_pers / gemm_w4)
int gemm_pers_init(hipStream_t s);                                 // creates the (current device, stream) ticket counters ahead of its first launch (optional)
int gemm_pers_set_cap(hipStream_t s, int cap);                       // at most `cap` workgroups per persistent launch on this stream (0 = #CUs)
void gemm_pers_forget(hipStream_t s);                               // drops one (idle) stream's counters before the stream is destroyed
void gemm_pers_release(int dev);                                   // frees every stream's counters of a device (last context on it destroyed)

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_SIGMOID = 2, ACT_GELU = 3 };

// ---- norms (norm.hip) ----
// y = LN(x) * w + b over the last dim D (D % 8 == 0, D <= 8192); rows independent; x,y bf16, stats fp32.
int launch_layernorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const bf16_t* w, const bf16_t* b,
                     int rows, int D, float eps, hipStream_t s, int silu = 0,   // silu = 1: y = SiLU(LN(x))
                     const bf16_t* res = nullptr, int ldres = 0);                // res: y = act(bf16(LN(x)) + res) (the RegNet block's tail; y may alias res)
int launch_rmsnorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const bf16_t* w, int rows, int D, float eps,
                   hipStream_t s);

// ---- ViT front end (vit.hip) ----
// frames [T,3,S,S] (bf16 or fp32) -> im2col patches A [T*G*G, Kpad] bf16 (k = c*P*P + py*P + px, zero padded)
// ---- patch_embed.hip (round 4): the ViT front end as one kernel, reading the frame tensor directly ----
bool patch_embed_supported(int S, int P, int D);                   // P in {14, 16}, D in {128, 256, 512, 1024}; other geometries keep the three-pass path below
size_t patch_embed_packed_elems(int P, int D);                     // elements of the repacked conv weight
int launch_patch_pack(const bf16_t* W, int ldw, bf16_t* wp, int D, int P, hipStream_t s);          // at load: [D][ldw] (k = c P P + ky P + j) -> fragment order, j padded to 16
int launch_cls_row(const bf16_t* cls, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb, bf16_t* out, int D, float eps, hipStream_t s);   // at load: the CLS row every frame gets
int launch_patch_embed(const void* frames, int frames_fp32, const bf16_t* wp, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb,
                       const bf16_t* cls_row, bf16_t* X, int T, int S, int P, int D, float eps, hipStream_t s);   // frames -> X = pre_layrnorm(cat(CLS, conv) + pos) [T (G G + 1), D]
int launch_im2col(const void* frames, int frames_fp32, bf16_t* A, int T, int S, int P, int Kpad, hipStream_t s);
// X[t, 0] = cls + pos[0];  X[t, 1+p] = PE[t*GG + p] + pos[1+p]      (PE = patch-embed GEMM output)
//   then X = LayerNorm(X) (pre_layrnorm), fused
int launch_vit_assemble(const bf16_t* PE, const bf16_t* cls, const bf16_t* pos, const bf16_t* lw, const bf16_t* lb,
                        bf16_t* X, int T, int GG, int D, float eps, hipStream_t s);

// ---- attention (attn.hip) ----
struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    long q_bs, q_hs; int q_rs;      // batch / head / row strides (elements)
    long k_bs, k_hs; int k_rs;
    long v_bs, v_hs; int v_rs;
    long o_bs, o_hs; int o_rs;
    int nq_rows;                    // query rows per (batch, head)
    int nkv_rows;                   // kv rows per (batch, kv head)
    int batch, heads, kv_heads;
    float scale;
    int causal;                     // 1: query row i attends kv rows <= i + (nkv_rows - nq_rows)
    // optional row-major view of V ([kv, hd] per (batch, kv head)): lets the head_dim-64 non-causal kernel fold a short tail of
    // keys (nkv % 64 <= 8) in on the VALU instead of running a mostly-masked tile; null = masked tile
    const bf16_t* Vrow; long vr_bs, vr_hs; int vr_rs;
    int dbg;                        // microbenchmark knock-outs (0 = full kernel)
    int v_perm;                     // 1: V^T was written by launch_transpose_v(..., perm = 1): selects the LDS-DMA ViT kernel;
                                    // 2: the same kernel reading V row-major through Vrow (LDS transpose reads) — V / transpose_v unused
};
// NB: V is passed PRE-TRANSPOSED: V^T[d, kv] with row stride v_rs (multiple of 64, >= nkv_rows, zero padded)
int launch_attn_vit(const AttnArgs& a, hipStream_t s);       // head_dim 64, non-causal, heads == kv_heads
int launch_attn_prefill(const AttnArgs& a, hipStream_t s);   // head_dim 128, causal, GQA 4:1
// V [n, hd] (row stride src_rs) -> V^T [hd, dst_rs] per (batch, head); zero fill beyond n
int launch_transpose_v(const bf16_t* src, long src_bs, long src_hs, int src_rs, bf16_t* dst, long dst_bs, long dst_hs,
                       int dst_rs, int n, int hd, int heads, int batch, hipStream_t s, int perm = 0);
// true: launch_attn_vit can run its LDS-DMA kernel for this key count -> transpose V with perm = 1 and set AttnArgs::v_perm
bool attn_vit_wants_perm(int nkv_rows, bool has_vrow);
// true: with attn_vit_wants_perm(), set v_perm = 2 and skip launch_transpose_v altogether (false only under an A/B debug switch)
bool attn_vit_rowmajor_v();

// ---- SpatialSlotPool (slot_pool.hip) ----
// feats rows: frame t patch p at feats + (t*frame_stride + p*row_stride); out RES [T*S, D] bf16 (pre-readout)
// ws: launch_slot_pool_ws_floats(T, D) floats of scratch (per-part softmax partials)
int launch_slot_pool(const bf16_t* feats, long frame_stride, int row_stride, const bf16_t* ln_w, const bf16_t* ln_b,
                     const bf16_t* slots /*[D,S]*/, const float* cos_t, const float* sin_t /*[n,D/2]*/,
                     bf16_t* res, int T, int n, int D, int S, float eps, float* ws, size_t ws_floats, hipStream_t s);
size_t launch_slot_pool_ws_floats(int T, int D);

// ---- LLM glue (llm.hip) ----
struct GatherTabs { const bf16_t* t[6]; };
// out[r, :] = tabs.t[kind[r]][row[r], :]   (embedding splice / per-frame [slots | time tokens] interleave)
int launch_gather_rows(const GatherTabs& tabs, const int32_t* kind, const int32_t* row, bf16_t* out, int L, int H,
                       hipStream_t s);
// RoPE on q (in place) and k (-> cache at pos) + v copy (-> cache).  qkv rows are [q heads | k heads | v heads].
// Row r: pos = pos_arr ? pos_arr[r] : pos0 + r ; slot = slot_arr ? slot_arr[r] : slot0.
int launch_rope_kv(bf16_t* qkv, int ld, bf16_t* kcache, bf16_t* vcache, long slot_stride, long kv_head_stride,
                   const int32_t* slot_arr, const int32_t* pos_arr, int slot0, int pos0, int R, int nq, int nkv, int hd,
                   const float* cos_t, const float* sin_t, int seq_len, hipStream_t s);   // seq_len > 0: R/seq_len prefill sequences end to end -> slots slot0, slot0+1, ...

// ---- decode (decode.hip) ----
// out[b, n] = sum_k X[b,k] W[n,k]  (B <= 64) (+ residual / SwiGLU on interleaved W)
// tiled: W is the decode copy made by launch_tile_pack ([N/16][K/64][64][16]); else row-major [N][ldw].
// ws / tickets: workspace of skinny_ws_floats(N, K, epi) floats and >= N/16 zero-initialised tickets.
// EPI_PARTIAL: no bf16 output; fp32 partial rows [skinny_ks(N, K, epi, B)][SK_ROWS][N] in ws, summed by launch_add_rmsnorm.
int launch_skinny_gemm(const bf16_t* X, int ldx, const bf16_t* W, int ldw, bf16_t* out, int ldo, const bf16_t* R,
                       int ldr, int B, int N, int K, int epi, int tiled, float* ws, size_t ws_floats, unsigned int* tickets,
                       int ntickets, hipStream_t s);
// the same GEMV fed by the PREVIOUS GEMV's partial rows: sums them, adds the residual R (-> new residual xout, != R), RMS-normalises (weight w) and parks the
// result as its activations — add_rmsnorm folded into the next GEMV (B <= 4; decode.hip SkinnyPro); output = EPI_PARTIAL rows in ws (!= part_in)
int launch_skinny_gemm_fused_norm(const float* part_in, int ks_in, const bf16_t* R, int ldr, bf16_t* xout, int ldx, const bf16_t* w, float eps,
                                  const bf16_t* Wtiled, int B, int N, int K, float* ws, size_t ws_floats, hipStream_t s);
// the down GEMV fed by the gate|up GEMV's partial rows [ks_gu][SK_ROWS][2 K]: SwiGLU (swiglu_combine's arithmetic) folded into its parking step (B <= 4)
int launch_skinny_gemm_fused_swiglu(const float* part_gu, int ks_gu, const bf16_t* Wtiled, int B, int N, int K, float* ws, size_t ws_floats, hipStream_t s);
bool skinny_fused_norm_ok(int N, int K, int B);      // false: this shape keeps the GEMV + add_rmsnorm pair
size_t skinny_ws_floats(int N, int K, int epi);
int skinny_ks(int N, int K, int epi, int B);
int launch_tile_pack(const bf16_t* src, int ldw, bf16_t* dst, int N, int K, hipStream_t s);
// out[b][j] = bf16(silu(sum_ks gate) * sum_ks up) over EPI_PARTIAL rows [KS][SK_ROWS][N2] of the 16-row interleaved gate|up product
int launch_swiglu_combine(const float* part, int KS, int N2, bf16_t* out, int ldo, int B, hipStream_t s);
// x = bf16(sum_ks part[ks][b]) + R[b] -> xout (may alias R); y = RMSNorm(x) * w.  N <= 4096.
// y8 / sy (optional): y also as e4m3 [B][N] + per-row scale, as launch_quant_rows_fp8(y) would give (fp8 weight path)
int launch_add_rmsnorm(const float* part, int KS, const bf16_t* R, int ldr, bf16_t* xout, int ldx, const bf16_t* w, bf16_t* y,
                       int ldy, int B, int N, float eps, hipStream_t s, uint8_t* y8 = nullptr, float* sy = nullptr);
// single-query GQA attention over the cache (context = pos[b] + 1 rows, split nsplit ways, <= 128 rows per split).
// fuse_rope = 1: `qkv` rows are the raw [q | k | v] projections of the new token: q and the new k are rotated here
// (RoPE at pos[b]) and k/v appended to the cache at row pos[b].  fuse_rope = 0: `qkv` holds ready q rows (ld ldq)
// and the cache already contains row pos[b].  tickets: zeroed uint32 [B*nkv].  O [B, nq*hd].
// the fused prologue of launch_attn_decode(fuse_rope = 1, qpart) as a kernel of its own: partial rows -> roped bf16 q rows in qout, k / v appended
int launch_qkv_finish(const float* part, int ks, int ldq, bf16_t* qout, bf16_t* kcache, bf16_t* vtcache, long slot_stride, long kv_head_stride,
                      int ctx_stride, const int32_t* slots, const int32_t* pos, int B, int nq, int nkv, const float* cos_t, const float* sin_t,
                      hipStream_t s);
int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* kcache, bf16_t* vtcache, long slot_stride, long kv_head_stride,
                       int ctx_stride, const int32_t* slots, const int32_t* pos, bf16_t* O, int ldo, float* ws, unsigned int* tickets, int B,
                       int nq, int nkv, int hd, int nsplit, float scale, int fuse_rope, const float* cos_t, const float* sin_t,
                       const float* qpart, int qks, hipStream_t s);   // qpart: fp32 partial rows [qks][SK_ROWS][ldq] instead of bf16 qkv
// heads: logits over [text V+1 | time Tv | score Sv] rows of Wh [NV_pad, H]; only tiles intersecting an active
// head's range are computed.  part: [B, ntiles] (max,idx).  logits_out optional [B, NV] fp32 (masked -inf).
int launch_head_logits(const bf16_t* X, int ldx, const bf16_t* Wh, int H, const int32_t* heads, int V, int Tv, int Sv,
                       float* part_val, int32_t* part_idx, float* logits_out, int B, hipStream_t s);
// argmax over partials + state machine + next-token embedding
struct StepState {
    int32_t* heads;       // [B] in/out
    int32_t* pos;         // [B] in/out (position of the NEXT token to be written)
    int32_t* done;        // [B] in/out
    int32_t* out_ids;     // [B, max_new]
    int32_t* out_len;     // [B]
    int32_t* step;        // [1] device step counter
    const int32_t* forced;// [B, max_new] teacher-forcing ids; negative entry = feed the arg-max
    const int32_t* params;// device [3]: max_new, eos, record_feed (device-resident so a captured graph stays valid)
};
int launch_select_next(const float* part_val, const int32_t* part_idx, const StepState& st, const bf16_t* embed,
                       const bf16_t* time_tab, const bf16_t* score_tab, const bf16_t* sync_row, bf16_t* xnext, int ldx,
                       int B, int H, int V, int Tv, int Sv, int advance, hipStream_t s);

// ---- fp8 (e4m3) weight path of the decoder (fp8.hip; the fp8 GEMM is launch_gemm_bf16 with GemmArgs::fp8 set) ----
// X bf16 [rows][K] -> X8 e4m3 [rows][K] + sx[row] = amax/448 (per-row dynamic scale).  Also used row-wise on weight matrices at load.
int launch_quant_rows_fp8(const bf16_t* X, long ldx, uint8_t* X8, long ld8, float* sx, int rows, int K, hipStream_t s);
// W8 [N][K] bytes -> decode copy [N/16][K/128][64 lanes][32 B]
int launch_tile_pack_fp8(const uint8_t* src, long ldw, uint8_t* dst, int N, int K, hipStream_t s);
// decode GEMV on fp8 operands: fp32 partial rows [skinny_fp8_ks(N,K,B)][SK_ROWS][N] = (X8 . W8^T) * sx[m] * sw[n] in ws
int launch_skinny_fp8(const uint8_t* X8, long ldx, const float* sx, const uint8_t* Wtiled, const float* sw, int B, int N, int K, float* ws,
                      size_t ws_floats, hipStream_t s);
int skinny_fp8_ks(int N, int K, int B);
// weight-only (W8A16) form of the same GEMV: bf16 activations, e4m3 weights widened to bf16 in registers, bf16 MFMA; partial rows
// [skinny_w8_ks(N,K,B)][SK_ROWS][N] = (X . W8^T) * sw[n]
int launch_skinny_w8(const bf16_t* X, long ldx, const uint8_t* Wtiled, const float* sw, int B, int N, int K, float* ws, size_t ws_floats,
                     hipStream_t s);
int skinny_w8_ks(int N, int K, int B);

// ---- STC connector support (stc.hip): channels-last [n][h][w][C] row kernels ----
int launch_dwconv3x3(const bf16_t* x, const bf16_t* w /*[C][9]*/, bf16_t* y, int N, int H, int W, int C, hipStream_t s);
int launch_avgpool(const bf16_t* x, bf16_t* y /*[N][C]*/, int N, int HW, int C, hipStream_t s);
int launch_bias_act(bf16_t* x, const bf16_t* bias, long rows, int C, int act, hipStream_t s);
int launch_scale_rows(bf16_t* x, const bf16_t* gate /*[N][C]*/, int N, int HW, int C, hipStream_t s);
int launch_add_act(bf16_t* x, const bf16_t* y, long n, int act, hipStream_t s);
int launch_im2col3d(const bf16_t* x, bf16_t* A, int T, int H, int W, int C, int To, int Ho, int Wo, hipStream_t s);
int launch_permute_conv3d_w(const bf16_t* w, bf16_t* out, int Co, int Ci, hipStream_t s);

// ---- frame preprocessing (preproc.hip): Pillow 8-bit bicubic resample passes + rescale/normalise table ----
int launch_resize_h(const uint8_t* frames, int T, int H, int W, int y0, int x0, uint32_t bg, const int32_t* bounds,
                    const int32_t* kk, int ksize, int row_first, int nrows, int S, uint8_t* tmp, hipStream_t s);
int launch_resize_v_norm(const uint8_t* tmp, int T, int nrows, int S, const int32_t* bounds, const int32_t* kk, int ksize,
                         const float* lut, void* out, int out_f32, hipStream_t s);
