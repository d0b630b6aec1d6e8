/* libtrace_hip.so — C ABI of the MI355X-native TRACE inference hot path.
 *
 * The reference (gyxxyg/TRACE) is 100% Python and has no FFI of its own; its arithmetic lives in
 * transformers/torch.  This header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md):
 * everything from the sampled frame tensor to the greedy token ids.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference repo root).
 *
 * Conventions: extern "C"; every function returns 0 on success or a negative TRACE_ERR_* code, the message is
 * available from trace_last_error(); no C++ exceptions cross the ABI; pointers are raw device (or, where
 * stated, host) addresses + explicit sizes; `stream` is a hipStream_t passed as void* (0 = default stream).
 * A trace_ctx owns its weights, KV cache and workspaces (hipMalloc) and is not thread-safe; one per process/GPU.
 * All device tensors are bf16 (uint16 bit patterns) unless stated.
 */
#ifndef TRACE_HIP_H
#define TRACE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TRACE_ABI_VERSION 4

typedef struct trace_ctx trace_ctx;

/* Geometry: the config.json keys the reference reads (trace/model/language_model/trace_mistral.py:84-96,
 * trace/model/multimodal_encoder/clip_encoder.py:15-16, trace/model/multimodal_projector/builder.py:413-421). */
typedef struct trace_config {
    int32_t vocab_size, hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads;
    int32_t time_vocab, score_vocab;
    float rms_eps, rope_theta;
    int32_t v_hidden, v_inter, v_layers_used, v_heads, v_image, v_patch;
    float v_eps;
    int32_t num_slots;
    float slot_eps, slot_rope_base;
    int32_t max_frames;      /* largest T per video                                    */
    int32_t max_ctx;         /* KV-cache length per sequence slot (prefill + new tokens) */
    int32_t max_batch;       /* KV-cache sequence slots, <= 512; one decode batch takes at most 256 of them (64 on the fp8 path; the rest can be
                              * prefilled meanwhile: trace_amd.engine.TraceEngine.generate_stream) */
    int32_t max_new_tokens;  /* capacity of the on-device output id buffer               */
    int32_t projector_type;  /* 0 = spatial_slot (TRACE), 1 = stc_connector (legacy API) */
    int32_t vit_batch_frames; /* frames one trace_vit_forward call may take (ViT workspaces); 0 = max_frames.  Larger than
                               * max_frames lets a caller push the frames of several videos through the tower together (the
                               * tower is per-frame: results do not depend on the grouping) */
    int32_t llm_weights_fp8;  /* the four decoder projections per layer on the fp8 path (BASELINE config 5): e4m3 weights with a scale per output
                               * row made at load from the bf16 tensors.  1 = W8A8 everywhere: activations quantised per token row on the
                               * fly, fp8 MFMA with fp32 accumulation.  2 = W8A8 prefill GEMMs, WEIGHT-ONLY decode GEMVs (bf16 activations,
                               * weights widened to bf16 in registers, bf16 MFMA): the same weight bytes per step, half the rounding-noise
                               * variance where the tokens are chosen.  No reference counterpart; parity anchor = the bf16 path */
} trace_config;

const char* trace_last_error(void);
int trace_abi_version(void);
/* The library's 16-bit element type: 0 = bf16 (libtrace_hip.so), 1 = IEEE fp16 (libtrace_hip_f16.so: the same sources compiled with -DTRACE_F16 —
 * the reference's own inference dtype, torch.float16, trace/model/builder.py:50,127,147 / trace/eval/evaluate.py:316).  Every "bf16" in this header
 * reads "the library's element type"; the two libraries export the same symbols and may be loaded side by side (RTLD_LOCAL). */
int trace_element_type(void);

/* from_pretrained (trace/model/builder.py:113-114): create, stream tensors in by their reference state-dict
 * names (bf16; `on_device` says whether `data` is a device or host pointer), then finalize. */
int trace_ctx_create(const trace_config* cfg, int device_id, trace_ctx** out);
int trace_ctx_destroy(trace_ctx* ctx);
int trace_ctx_load_tensor(trace_ctx* ctx, const char* name, const void* data, int on_device, const int64_t* shape,
                          int ndim);
int trace_ctx_finalize(trace_ctx* ctx);
int64_t trace_ctx_device_bytes(trace_ctx* ctx);

/* Frame preprocessing of process_video (trace/mm_utils.py:456-462; expand2square :259-270; HF CLIPImageProcessor.preprocess of
 * transformers 4.40.1: resize shortest edge -> v_image with Pillow BICUBIC, centre crop, x/255, (x - mean)/std), on the device:
 * frames_u8 [T,H,W,3] uint8 RGB (device) -> out [T,3,v_image,v_image] (out_dtype 0 = bf16, 1 = fp32; device).
 * pad_to_square = the drivers' aspect_ratio == 'pad' (background int(mean*255)).  image_mean / image_std: 3 host floats.
 * The resize reproduces Pillow's 8-bit two-pass fixed-point resampler bit for bit; fp32 output equals the reference's
 * FloatTensor exactly. */
int trace_preprocess_frames(trace_ctx* ctx, const void* frames_u8, int T, int H, int W, int pad_to_square,
                            const float* image_mean, const float* image_std, void* out, int out_dtype, void* stream);

/* CLIPVisionTower.forward + feature_select (trace/model/multimodal_encoder/clip_encoder.py:31-53):
 * frames [T,3,S,S] (dtype 0 = bf16, 1 = fp32, device) -> features [T, patches, v_hidden] bf16 = hidden state
 * after encoder layer v_layers_used, CLS dropped.  feats_out may be NULL (kept internally for trace_slot_pool). */
int trace_vit_forward(trace_ctx* ctx, const void* frames, int frames_dtype, int T, void* feats_out, void* stream);

/* SpatialSlotPool.forward (trace/model/multimodal_projector/builder.py:427-467): feats (NULL = the internal
 * buffer of the last trace_vit_forward) -> slots_out [T, num_slots, hidden]; slots_out may be NULL. */
int trace_slot_pool(trace_ctx* ctx, const void* feats, int T, void* slots_out, void* stream);

/* STCConnector.forward (trace/model/multimodal_projector/builder.py:208-249), legacy trace.infer() path only
 * (projector_type = 1): feats [T, patches, v_hidden] (NULL = internal buffer of the last trace_vit_forward) ->
 * out [(T/2+1) * (g/2+1)^2, hidden] (may be NULL; the result also becomes the internal video rows for
 * trace_splice_embeds).  *rows_out receives the row count.  Parity of this connector is unpinned (timm absent). */
int trace_stc_connector(trace_ctx* ctx, const void* feats, int T, void* out, int* rows_out, void* stream);

/* encode_images_or_videos (trace/model/trace_arch.py:218-266): ViT + slot pool + per-frame time-token embedding;
 * time_ids is HOST int32 [T, 6] (TimeTower.encode(t)[:-1]).  Result [T*(slots+6), hidden] stays internal;
 * video_out (device, may be NULL) receives a copy. */
int trace_encode_video(trace_ctx* ctx, const void* frames, int frames_dtype, int T, const int32_t* time_ids,
                       void* video_out, void* stream);

/* The same from ViT features computed earlier (trace_vit_forward with feats_out, possibly in a call that carried the frames
 * of several videos, see vit_batch_frames): feats [T, patches, v_hidden] bf16 (device) -> slot pool + time-token rows. */
int trace_encode_features(trace_ctx* ctx, const void* feats, int T, const int32_t* time_ids, void* video_out, void* stream);

/* prepare_inputs_labels_for_multimodal, prefill branch (trace/model/trace_arch.py:377-456): HOST ids with the
 * modal placeholders (-201 video, -203 time, -204 score, -205 sync); the single video placeholder expands to
 * the rows of the last trace_encode_video.  time_rows/score_rows: HOST tower row ids consumed in order by the
 * -203/-204 placeholders (may be NULL).  Writes the spliced embeddings internally, returns their length in
 * *L_out; embeds_out (device [L,hidden], may be NULL) receives a copy. */
int trace_splice_embeds(trace_ctx* ctx, const int32_t* ids, int n_ids, const int32_t* time_rows, int n_time,
                        const int32_t* score_rows, int n_score, int* L_out, void* embeds_out, void* stream);

/* TraceMistralForCausalLM.forward, prefill (trace/model/language_model/trace_mistral.py:114-264): runs the L
 * spliced rows (embeds == NULL: internal buffer) through the decoder into KV slot `slot`.  hidden_out (device
 * [L,hidden] bf16, may be NULL) receives the final-norm hidden states (tests). */
int trace_llm_prefill(trace_ctx* ctx, int slot, const void* embeds, int L, void* hidden_out, void* stream);
/* Two prompts of EQUAL spliced length prefilled in one pass (GEMM M = 2L fills the MFMA tile grid in whole rounds): embeds0 / embeds1
 * [L, hidden] bf16 device (the embeds_out of two trace_splice_embeds calls) -> KV slots slot0 and slot0 + 1.  Same results as two
 * trace_llm_prefill calls. */
int trace_llm_prefill_pair(trace_ctx* ctx, int slot0, const void* embeds0, const void* embeds1, int L, void* stream);
/* The same for n <= 8 prompts of equal length while n * L <= max(4 * max_ctx, min(8192, 8 * max_ctx)) rows (the prefill workspaces; more -> TRACE_ERR_ARG):
 * embeds = HOST array of n device pointers ([L, hidden] bf16 each) -> KV slots slot0 .. slot0 + n - 1.  M = 4 L = 7868 at the C2 shape fills the 256x256
 * tile grid of every projection in whole rounds; at the C4 shape (L = 1086) seven prompts do (30 row panels: the o / down grid in 1.9 rounds where four
 * prompts' 17 panels need 2 rounds for 1.06 of work).  Results per prompt do not depend on n. */
int trace_llm_prefill_multi(trace_ctx* ctx, int slot0, const void* const* embeds, int n, int L, void* stream);

/* The head stage of forward() for EVERY position (trace_mistral.py:190-252: lm_head | sync_head | time_head | score_head, fp32,
 * everything outside head `head`'s id range set to -inf): hidden [R, hidden] bf16 device = the hidden_out of trace_llm_prefill ->
 * logits_out [R, V+1+Tv+Sv] fp32 device.  The decode loop needs the last row only and gets it from trace_decode_begin / _steps. */
int trace_llm_head_logits(trace_ctx* ctx, const void* hidden, int R, int head, float* logits_out, void* stream);

/* generate() = greedy loop with head switching (trace_mistral.py:268-347 + HF greedy search).
 * begin: sequences = the given KV slots (each prefilled); heads[b] in {0 text,1 time,2 score} (callers pass [1]);
 *        computes token 0 from the prefill hidden state.  forced: HOST [B, max_new] teacher-forcing ids or NULL.
 *        eos < 0 disables the stop.  logits_out (device fp32 [B, V+1+Tv+Sv], may be NULL) = masked logits of step 0.
 * steps: runs n more decode steps entirely on device (use_graph: hipGraph replay; logits_out only with n == 1).  At most
 *        max_new - 1 steps in total after one begin (every step appends a KV row; TRACE_ERR_STATE beyond).
 * read : synchronises and copies ids [B, max_new] / lengths [B] / current heads [B] to HOST buffers. */
int trace_decode_begin(trace_ctx* ctx, const int32_t* slots, int B, const int32_t* heads, int max_new, int eos,
                       const int32_t* forced, float* logits_out, void* stream);
int trace_decode_steps(trace_ctx* ctx, int n, int use_graph, float* logits_out, void* stream);
int trace_decode_read(trace_ctx* ctx, int32_t* out_ids, int32_t* out_len, int32_t* heads, void* stream);
/* Host-driven token selection (do_sample=True in scripts/inference/inference.py:62, stopping criteria): with host
 * mode on (set before trace_decode_begin), begin/steps stop after the masked head logits; the host picks the ids and
 * trace_decode_feed applies them (output record, head switch, next-token embedding).  One eager step at a time. */
int trace_decode_host_mode(trace_ctx* ctx, int on);
int trace_decode_feed(trace_ctx* ctx, const int32_t* tokens, int B, void* stream);

/* Two-stage pipeline support (trace/eval/evaluate.py:298-417 loops over independent videos: while one batch decodes — HBM-bound —
 * the next batch's ViT + prefill — MFMA-bound — can run on another stream into other KV slots; the stages share no buffers).
 * trace_stream_create: a HIP stream confined to cu_count CUs starting at logical CU cu_first (hipExtStreamCreateWithCUMask; mask bit i
 * is CU i / 8 of XCD i % 8, so a run of bits is spread evenly over the XCDs; cu_first and cu_count multiples of 8); cu_count == 0: an
 * ordinary non-blocking stream.  A CU-masked stream's persistent GEMMs launch at most cu_count workgroups (one per CU the stream can use);
 * trace_set_gemm_cus overrides that number for the streams this context has created (n workgroups; 0 = the device's CU count) — it is a
 * property of those streams and goes away with them.  Streams are destroyed with trace_stream_destroy (idle). */
int trace_stream_create(trace_ctx* ctx, int cu_first, int cu_count, void** stream_out);
int trace_stream_destroy(trace_ctx* ctx, void* stream);
int trace_set_gemm_cus(trace_ctx* ctx, int n);

/* Timing hook for bench.py: average device time (ms, hipEvents on `stream`) of the last trace_decode_steps call
 * per step, and of its skinny-GEMM launches if profiling was enabled with trace_set_profile(ctx, 1). */
int trace_set_profile(trace_ctx* ctx, int on);
/* Debugging aid: device addresses of the K cache, the V^T cache and the prefill's last-position hidden rows, with strides[8] = layer, slot, kv-head
 * strides (elements), ctx_pad, layers, kv heads, head_dim, hidden (tools/pipeline_stress.py checksums them between the pipeline's stages). */
int trace_debug_buffers(trace_ctx* ctx, void** kcache, void** vcache, void** xlast, int64_t* strides);
/* out[0..n) (n <= 20): [0] ms per decode step of the last trace_decode_steps call, [1] its steps, [2] average ms of the bracketed decode launch,
 * [3] its samples, [4] its algorithmic bytes, [5] average ms of the bracketed ViT fc1 GEMM launch, [6] its samples, [7] its GFLOP,
 * [8] which decode launch took the bracket: 1 = gate|up GEMV, 2 = batch-1 fused-norm gate|up GEMV, 3 = the wide step's layer-0 decode attention,
 * [9] always 0 (rounds 3-4: whether the bracketed ViT GEMMs ran with the LayerNorm fold; the fold left the product in round 5),
 * [12..14] average ms of the bracketed ViT qkv / out-proj / fc2 GEMM launches (layer 0, the same calls as [5]), [15..17] their GFLOP. */
int trace_get_profile(trace_ctx* ctx, float* out, int n);
/* Which per-launch brackets profiling mode 2 takes: bit 0 = the ViT fc1 GEMM, bit 1 = the decode step's dominant kernel.  A pipelined caller
 * (two stages on two streams) leaves a stage's bracket on only while that stage has the GPU to itself (pipeline fill / drain). */
int trace_set_profile_brackets(trace_ctx* ctx, int mask);

/* ---- kernel-level entry points (unit tests / microbenchmarks; raw device pointers) ---- */
int trace_op_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* R,
                  int ldr, int M, int N, int K, int epilogue, void* stream);
/* GEMM kernel selection for tests / microbenchmarks (process-wide): 0 auto, 2 = 128^2 tiles, 3 = 256^2 tiles (gemm.hip), 4 = gemm_ldr.hip, 5-7 = gemm_pers.hip
 * (ticketed / static deal / one tile per workgroup), 8 = gemm_w4.hip.  Other ranges are A/B switches of single kernels (engine.hip trace_op_set_gemm_variant:
 * 300 + opt K-loop builds, 500-501 tile walk, 520-521 residual shapes on the persistent kernel, 530-531 gemm_pers / gemm_w4 in auto mode, 540 + opt gemm_w4 builds, 190-192 ViT attention ...). */
int trace_op_set_gemm_variant(int variant);
/* profiling: device buffer of 8 x uint64 per workgroup receiving phase time stamps of every later GEMM launch (NULL = off) */
int trace_op_set_gemm_trace(void* buf);
int trace_op_layernorm(const void* x, void* y, const void* w, const void* b, int rows, int D, float eps, void* stream);
int trace_op_rmsnorm(const void* x, void* y, const void* w, int rows, int D, float eps, void* stream);
int trace_op_attention(const void* Q, const void* K, const void* V, void* O, void* vt_scratch, int batch, int heads,
                       int kv_heads, int nq, int nkv, int head_dim, int causal, float scale, void* stream);
/* Decode GEMV out[b,n] = sum_k X[b,k] W[n,k], B <= 64.  w_tiled: W in the decode tile layout written by
   trace_op_tile_pack ([N/16][K/64][64][16]) instead of row-major.  epilogue 0 none, 1 +R, 3 SwiGLU (16-row interleaved
   gate|up), 4 partial: `out` = fp32 k-chunk partial rows [trace_op_skinny_ks()][trace_op_sk_rows()][N] for trace_op_add_rmsnorm. */
int trace_op_skinny_gemm(const void* X, const void* W, void* out, const void* R, int B, int N, int K, int epilogue,
                         int w_tiled, void* stream);
int trace_op_skinny_ks(int N, int K, int epilogue, int B);
/* Decode GEMV of 1..4 rows with the preceding "sum the partial rows + residual -> new residual, RMSNorm" folded in: part_in [ks_in][sk_rows][K] fp32
   (ks_in may be 0) + R [B,K] -> xout [B,K]; out = fp32 partial rows [trace_op_skinny_ks(N,K,4,B)][sk_rows][N] of RMSNorm(xout; w, eps) . W^T */
int trace_op_skinny_fused_norm(const float* part_in, int ks_in, const void* R, void* xout, const void* w, float eps, const void* W, float* out,
                               int B, int N, int K, void* stream);
int trace_op_sk_rows(void);                 /* row stride of every fp32 partial-row buffer = the largest decode batch (128) */
/* Decode batches above 64 rows: out = X[M <= 128, K] . W[N, K]^T (row-major W) as fp32 k-chunk partial rows
   [trace_op_gemm_partial_ks(N, K)][trace_op_sk_rows()][N] for trace_op_add_rmsnorm (split-K MFMA GEMM, 128x128 tiles) */
int trace_op_gemm_partial_ks(int N, int K);
int trace_op_gemm_partial(const void* A, const void* W, float* part, int M, int N, int K, int w_tiled, void* stream);
/* w_tiled: W is the trace_op_tile_pack copy (1; 5 = with the 4-stage K-tile ring the engine uses).  The gate|up product of such a step: tiled 16-row interleaved
   gate|up matrix, SwiGLU epilogue, out [M, N/2] bf16 */
int trace_op_gemm_swiglu_tiled(const void* X, const void* Wt, void* out, int M, int N, int K, int ring, void* stream);
int trace_op_tile_pack(const void* W, void* Wt, int N, int K, void* stream);
/* fp8 path pieces: row quantiser (X bf16 [rows,K] -> e4m3 bytes + scale[row] = amax/448), the W8A8 GEMM
   C = (A8 . W8^T) * sa[m] * sw[n] (+ residual / SwiGLU epilogue as trace_op_gemm), and the decode GEMV (fp32 out [B,N]) */
int trace_op_quant_rows_fp8(const void* X, void* X8, float* sx, int rows, int K, void* stream);
int trace_op_gemm_fp8(const void* A8, const float* sa, const void* W8, const float* sw, void* C, const void* R, int M, int N, int K,
                      int epilogue, void* stream);
int trace_op_skinny_fp8(const void* X8, const float* sx, const void* W8, const float* sw, float* out, int B, int N, int K, void* stream);
/* the weight-only decode GEMV: X bf16 [B,K] . (e4m3 W8 [N,K] widened to bf16)^T * sw[n] -> fp32 out [B,N] */
int trace_op_skinny_w8(const void* X, const void* W8, const float* sw, float* out, int B, int N, int K, void* stream);
/* out[b][j] = bf16(silu(sum_ks gate) * sum_ks up) from partial rows [KS][64][N2] of the 16-row interleaved gate|up GEMV */
int trace_op_swiglu_combine(const float* part, int KS, int N2, void* out, int B, void* stream);
/* x = bf16(sum_ks part[ks][b][:]) + R[b][:] -> xout;  y = RMSNorm(x) * w   (decode residual add + norm, N <= 4096) */
int trace_op_add_rmsnorm(const float* part, int KS, const void* R, void* xout, const void* w, void* y, int B, int N,
                         float eps, void* stream);
/* kcache [B, nkv, max_ctx, 128] row-major; vtcache [B, nkv, 128, max_ctx] = V transposed (the engine's cache layout,
   max_ctx % 32 == 0); pos[b] = newest position, already in both caches; q [B, nq*128] rotated; ws B*nq*nsplit*130 floats.
   The kernel fetches whole 32-position blocks: positions pos[b] + 1 .. the next multiple of 32 are READ (their scores are masked
   and their weights are exactly 0, so what they hold never counts) — both caches must hold finite values there (the engine's
   caches are zero-filled at creation and only ever hold finite values). */
int trace_op_attn_decode(const void* q, const void* kcache, const void* vtcache, const int32_t* pos, void* O, float* ws,
                         int B, int nq, int nkv, int max_ctx, int nsplit, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
