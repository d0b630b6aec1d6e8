// Engine behind include/trace_hip.h: owns weights (repacked for the kernels), KV cache and workspaces, and
// sequences the HIP kernels for the three phases of the path
//   video encode (CLIP ViT -> SpatialSlotPool -> [slots | time tokens])  -> splice -> Mistral prefill
//   -> on-device greedy decode with head switching (hipGraph-replayed step, no host round trip per token).
// One context per process / GPU; videos are independent, so multi-GPU is one context per rank (trace_amd/dist.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/trace_hip.h"
#include "common.h"
#include "kernels.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(TRACE_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define LCHK(x) do { int r_ = (x); if (r_ != TRACE_OK) return fail(r_, std::string("launch failed: ") + #x); } while (0)
#define TRY(x) do { int r_ = (x); if (r_ != TRACE_OK) { if (g_err.empty()) g_err = std::string("failed: ") + #x; return r_; } } while (0)

struct VitLayer {
    bf16_t *ln1w, *ln1b, *wqkv, *bqkv, *wo, *bo, *ln2w, *ln2b, *w1, *b1, *w2, *b2;
};
struct LlmLayer {
    bf16_t *rms1, *wqkv, *wo, *rms2, *wgu, *wd;
    bf16_t *wqkv_d, *wo_d, *wgu_d, *wd_d;      // decode copies in the GEMV tile layout (decode.hip: launch_tile_pack)
    // fp8 weight path (trace_config::llm_weights_fp8; fp8.hip): e4m3 row-major copies for the prefill GEMMs, tile-layout copies for the
    // decode GEMVs, one fp32 scale per output row
    uint8_t *wqkv8, *wo8, *wgu8, *wd8, *wqkv8_d, *wo8_d, *wgu8_d, *wd8_d;
    float *sqkv, *so, *sgu, *sd;
};
struct StcBlock { bf16_t *w1, *n1w, *n1b, *wdw, *n2w, *n2b, *fc1w, *fc1b, *fc2w, *fc2b, *w3, *n3w, *n3b, *wd, *ndw, *ndb; int cin, rd; };

constexpr int PF_MAX = 4;        // equal-length prompts of max_ctx rows one prefill pass can take: the workspaces hold pf_rows() rows, and a pass takes up to PF_MAX_N
constexpr int PF_MAX_N = 8;      // SHORTER prompts while n x L fits them (round 6: four 1086-row Charades prompts are 17 row panels = 1.06 rounds of the o / down tile grid)
static size_t pf_rows(int max_ctx) { return std::max<size_t>((size_t)PF_MAX * max_ctx, std::min<size_t>(8192, (size_t)PF_MAX_N * max_ctx)); }
constexpr int MAX_SLOTS = 512;   // KV-cache sequence slots per context (a decode batch is at most SK_ROWS of them)
static int g_ctx_per_dev[16] = {0};

struct trace_ctx {
    trace_config c{};
    int dev = 0;
    // derived
    int H, I, NL, NQ, NKV, HD, V, Tv, Sv, NV, NVpad, QKV;
    int vh, vi, vL, vheads, G, GG, NT, P, Kpatch, Kpad, tokpad;
    int S, TPF;                      // slots per frame, tokens per frame (slots + 6)
    int max_ctx, max_B, ctx_pad;
    int vit_frames;                  // frames one trace_vit_forward call may take (>= max_frames)
    std::vector<void*> allocs;
    size_t total_bytes = 0;
    std::unordered_map<std::string, int> loaded;
    bool finalized = false;
    bool counted = false;            // in g_ctx_per_dev (the last context of a device frees the persistent GEMM's ticket counters)
    // weights
    bf16_t *patch_w, *cls, *pos_emb, *pre_w, *pre_b;
    bf16_t *patch_wp = nullptr, *cls_row = nullptr;      // fused patch embedding (patch_embed.hip): repacked conv weight, the CLS row
    std::vector<VitLayer> vit;
    bf16_t *sl_lnw, *sl_lnb, *sl_slots, *sl_readout;
    // STC connector (projector_type == 1)
    int stc = 0, stc_loaded = 0;
    StcBlock stc_blk[2][4];
    bf16_t *stc_w3d, *stc_b3d, *stc_r0w, *stc_r0b, *stc_r2w, *stc_r2b;
    bf16_t *stc_x, *stc_y, *stc_z, *stc_sc, *stc_col, *stc_pool, *stc_g1, *stc_g2, *stc_tmp;
    bf16_t *embed, *final_norm, *wheads, *time_tab, *score_tab, *sync_row;
    std::vector<LlmLayer> llm;
    float *slot_cos, *slot_sin, *rope_cos, *rope_sin;
    // KV cache: K [layer][slot][kvh][ctx_pad][hd] row-major; V TRANSPOSED [layer][slot][kvh][hd][ctx_pad] (decode.hip)
    bf16_t *kcache, *vcache;
    size_t kv_head_stride, slot_stride, layer_stride;
    // ViT workspaces
    bf16_t *vX, *vH, *vQKV, *vVT, *vMLP;
    bf16_t *sl_res, *sl_out, *video;     // [T*S, vh], [T*S, H], [T*TPF, H]
    float* sl_ws = nullptr; size_t sl_ws_floats = 0;   // slot pool: per-part softmax partials
    int video_rows = 0;
    // LLM prefill workspaces
    bf16_t *pX, *pH, *pQKV, *pO, *pACT;
    int32_t *d_kind, *d_row;             // splice index arrays (max_ctx)
    int32_t *h_kind, *h_row;             // pinned host staging: the buffer of the ring below that the current call fills
    static constexpr int NSTAGE_H = 4;   // ring of staging buffers, each behind an event recorded after its last copy was queued (round 6: the two calls that use
    int32_t* h_ring = nullptr;           // them waited for the whole stream instead — two drains of the encode stream per video)
    hipEvent_t h_ev[NSTAGE_H] = {nullptr, nullptr, nullptr, nullptr};
    int h_next = 0; size_t h_len = 0;
    int spliced_len = 0;
    // decode state
    bf16_t *dX, *dH, *dQKV, *dO, *dACT, *xlast;   // [16, *]
    float* attn_ws; unsigned int* tickets;
    void* pp_buf = nullptr; size_t pp_bytes = 0;       // frame preprocessing: tap tables + staged rows (grow-only)
    float* sk_ws = nullptr; unsigned int* sk_tickets = nullptr; size_t sk_ws_floats = 0; int sk_ntickets = 0;   // decode GEMV K-chunk partials
    float* sk_ws2 = nullptr;           // second partial-row buffer and residual rows: the fused-norm GEMVs of small batches read one and write the other
    bf16_t* dX2 = nullptr;
    float* part_val; int32_t* part_idx;
    float* hl_val; int32_t* hl_idx;        // trace_llm_head_logits' own partial buffers
    int32_t *d_slots, *d_pos, *d_heads, *d_done, *d_out_ids, *d_out_len, *d_step, *d_forced, *d_params;
    int32_t* d_heads_tmp;                // head id per row for trace_llm_head_logits
    int B = 0, max_new = 0, eos = -1, has_forced = 0, ntiles = 0, nsplit = 32;
    int slot_len[MAX_SLOTS] = {0};
    int fp8 = 0;                       // decoder projections on the fp8 path
    int fp8_wonly = 0;                 // llm_weights_fp8 == 2: the decode GEMVs keep bf16 activations (weight-only; the prefill GEMMs stay W8A8)
    uint8_t *pA8 = nullptr, *dA8 = nullptr, *dH8 = nullptr;      // quantised activations: prefill [2 max_ctx][max(H, I)], decode [64][I]; dH8 = the normed hidden rows
    float *psa = nullptr, *dsa = nullptr, *dsh = nullptr;        // their per-row scales
    int step_in_call = 0;              // decode steps taken since trace_decode_begin at the time decode_step runs (host copy)
    long pos_sum = 0;                  // sum over the batch of the prefill lengths (host copy: algorithmic KV bytes of a decode step's attention)
    double kbytes_sum = 0.0;           // algorithmic bytes of the bracketed launches (profile == 2)
    int host_mode = 0, fed = 0;        // host-driven token selection (sampling): head logits only, ids fed back by the host
    int steps_done = 0;                // decode steps taken since trace_decode_begin (bounded by max_new - 1: the KV slot and the RoPE tables end at max_ctx)
    hipGraphExec_t graphs[SK_ROWS + 1] = {nullptr};   // one captured decode step per batch size
    hipStream_t cap_stream = nullptr;
    std::vector<hipStream_t> streams;   // trace_stream_create
    // profiling
    int profile = 0;                  // 1: time decode_steps calls; 2: also bracket the layer-0 gate|up GEMV launch
    int bracket_mask = 3;             // which per-launch brackets profile == 2 takes: bit 0 = the ViT fc1 GEMM, bit 1 = the decode step's kernel
                                      // (trace_set_profile_brackets: a pipelined caller switches a stage's bracket off while the other stage's
                                      // kernels share the GPU with it — an event pair then times the queueing, not the kernel)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;      // event pool for per-launch brackets (eager mode)
    int kev_used = 0;
    hipEvent_t gev0 = nullptr, gev1 = nullptr;   // pair recorded from inside the captured graph
    double ksum_ms = 0.0; int ksamples = 0;
    hipEvent_t mev0 = nullptr, mev1 = nullptr;   // bracket of one ViT fc1 GEMM launch per trace_vit_forward (profile == 2)
    double msum_ms = 0.0; int msamples = 0; double mflops = 0.0; int mM = 0;
    // the other three GEMM shapes of the layer (qkv, out-proj, fc2), bracketed the same way in layer 0: the 256x256 MFMA GEMM family is the run's
    // dominant kernel and its four shapes sit at different fractions of the peak — the line reports each (pairs 0 = qkv, 1 = out-proj, 2 = fc2)
    hipEvent_t vev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double vsum_ms[3] = {0.0, 0.0, 0.0}; double vflops[3] = {0.0, 0.0, 0.0};
    float prof[20] = {0};
    int bracket_kind = 0;             // which launch the decode bracket of profile == 2 was put around: 0 none yet, 1 = gate|up GEMV (skinny path), 2 = the fused-norm
                                      // gate|up GEMV of a batch-1 step, 3 = the wide step's layer-0 decode attention
};

template <typename T>
static int dalloc(trace_ctx* c, T** p, size_t n_elems) {
    void* q = nullptr;
    const size_t bytes = (n_elems * sizeof(T) + 255) & ~(size_t)255;
    HIPCHK(hipMalloc(&q, bytes));
    HIPCHK(hipMemset(q, 0, bytes));
    c->allocs.push_back(q);
    c->total_bytes += bytes;
    *p = (T*)q;
    return TRACE_OK;
}

#define SKWS(c) (c)->sk_ws, (c)->sk_ws_floats, (c)->sk_tickets, (c)->sk_ntickets
static int round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" const char* trace_last_error(void) { return g_err.c_str(); }
extern "C" int trace_abi_version(void) { return TRACE_ABI_VERSION; }
extern "C" int trace_element_type(void) { return TRACE_ELEMENT_TYPE; }

extern "C" int trace_ctx_create(const trace_config* cfg, int device_id, trace_ctx** out) {
    if (!cfg || !out) return fail(TRACE_ERR_ARG, "null argument");
    g_err.clear();
    HIPCHK(hipSetDevice(device_id));
    trace_ctx* c = new trace_ctx();
    c->c = *cfg;
    c->dev = device_id;
    c->H = cfg->hidden_size; c->I = cfg->intermediate_size; c->NL = cfg->num_layers;
    c->NQ = cfg->num_heads; c->NKV = cfg->num_kv_heads; c->HD = c->H / c->NQ;
    c->V = cfg->vocab_size; c->Tv = cfg->time_vocab; c->Sv = cfg->score_vocab;
    c->NV = c->V + 1 + c->Tv + c->Sv; c->NVpad = round_up(c->NV, 16);
    c->QKV = (c->NQ + 2 * c->NKV) * c->HD;
    c->vh = cfg->v_hidden; c->vi = cfg->v_inter; c->vL = cfg->v_layers_used; c->vheads = cfg->v_heads;
    c->P = cfg->v_patch; c->G = cfg->v_image / cfg->v_patch; c->GG = c->G * c->G; c->NT = c->GG + 1;
    c->Kpatch = 3 * c->P * c->P; c->Kpad = round_up(c->Kpatch, 64); c->tokpad = round_up(c->NT, 64);
    c->S = cfg->num_slots; c->TPF = c->S + 6;
    c->max_ctx = cfg->max_ctx; c->max_B = cfg->max_batch; c->ctx_pad = round_up(c->max_ctx, 64);
    c->vit_frames = cfg->vit_batch_frames > cfg->max_frames ? cfg->vit_batch_frames : cfg->max_frames;
    auto bad = [&](const char* m) { delete c; return fail(TRACE_ERR_ARG, m); };
    if (c->HD != 128 || c->NQ != 4 * c->NKV) return bad("LLM kernels need head_dim 128 and 4:1 GQA");
    if (c->vh / c->vheads != 64) return bad("ViT kernels need head_dim 64");
    if (c->H % 128 || c->I % 128 || c->vh % 128 || c->vi % 128 || c->QKV % 128) return bad("dims must be multiples of 128");
    if (c->H > 4096) return bad("hidden_size > 4096: the decode row kernels (norms, residual add) hold one 4096-wide row per workgroup");
    c->stc = cfg->projector_type == 1;
    if (cfg->llm_weights_fp8 && TRACE_ELEMENT_TYPE != 0) { delete c; return fail(TRACE_ERR_ARG, "llm_weights_fp8 needs the bf16 library (the fp16 build has no fp8 weight path)"); }
    c->fp8 = cfg->llm_weights_fp8 != 0;
    c->fp8_wonly = cfg->llm_weights_fp8 == 2;
    if (c->fp8 && (c->H % 128 || c->I % 128 || c->I > 16384)) return bad("fp8 weight path needs hidden / intermediate sizes that are multiples of 128 (intermediate <= 16384)");
    if (!c->stc && (c->S != 8 || c->vh > 1024)) return bad("slot pool kernel needs 8 slots and mm_hidden <= 1024");
    if (c->max_B < 1 || c->max_B > MAX_SLOTS) return bad("max_batch (KV slots) must be in [1,512]");
    c->nsplit = 32;                                           // upper bound (workspace size); per-batch value below
    {
        const int g2 = c->G / 2 + 1;
        const int vis_rows = c->stc ? (cfg->max_frames / 2 + 1) * g2 * g2 : cfg->max_frames * c->TPF;
        if (cfg->max_frames < 1 || vis_rows > c->max_ctx) return bad("visual tokens of max_frames exceed max_ctx");
        if (c->stc && cfg->max_frames > 32) return bad("STC connector path supports at most 32 frames");
    }

    const size_t H = c->H, I = c->I, vh = c->vh, vi = c->vi;
    int rc = TRACE_OK;
#define A(p, n) if (rc == TRACE_OK) rc = dalloc(c, &(p), (size_t)(n))
    // --- weights ---
    A(c->patch_w, vh * c->Kpad); A(c->cls, vh); A(c->pos_emb, (size_t)c->NT * vh); A(c->pre_w, vh); A(c->pre_b, vh);
    if (patch_embed_supported(cfg->v_image, c->P, (int)vh)) { A(c->patch_wp, patch_embed_packed_elems(c->P, (int)vh)); A(c->cls_row, vh); }
    c->vit.resize(c->vL);
    for (auto& l : c->vit) {
        A(l.ln1w, vh); A(l.ln1b, vh); A(l.wqkv, 3 * vh * vh); A(l.bqkv, 3 * vh); A(l.wo, vh * vh); A(l.bo, vh);
        A(l.ln2w, vh); A(l.ln2b, vh); A(l.w1, vi * vh); A(l.b1, vi); A(l.w2, vh * vi); A(l.b2, vh);
    }
    if (!c->stc) { A(c->sl_lnw, vh); A(c->sl_lnb, vh); A(c->sl_slots, vh * c->S); A(c->sl_readout, H * vh); }
    else {
        for (int st = 0; st < 2; ++st)
            for (int b = 0; b < 4; ++b) {
                StcBlock& k = c->stc_blk[st][b];
                k.cin = (st == 0 && b == 0) ? (int)vh : (int)H;
                k.rd = (int)lround(k.cin * 0.25);
                A(k.w1, H * k.cin); A(k.n1w, H); A(k.n1b, H); A(k.wdw, H * 9); A(k.n2w, H); A(k.n2b, H);
                A(k.fc1w, (size_t)k.rd * H); A(k.fc1b, k.rd); A(k.fc2w, H * k.rd); A(k.fc2b, H);
                A(k.w3, H * H); A(k.n3w, H); A(k.n3b, H);
                if (k.cin != (int)H) { A(k.wd, H * k.cin); A(k.ndw, H); A(k.ndb, H); } else { k.wd = k.ndw = k.ndb = nullptr; }
            }
        A(c->stc_w3d, 8 * H * H); A(c->stc_b3d, H); A(c->stc_r0w, H * H); A(c->stc_r0b, H); A(c->stc_r2w, H * H); A(c->stc_r2b, H);
        const size_t rows = (size_t)cfg->max_frames * c->GG;
        const int g2 = c->G / 2 + 1;
        const size_t rows2 = (size_t)(cfg->max_frames / 2 + 1) * g2 * g2;
        A(c->stc_x, rows * H); A(c->stc_y, rows * H); A(c->stc_z, rows * H); A(c->stc_sc, rows * H);
        A(c->stc_col, rows2 * 8 * H); A(c->stc_pool, 32 * H); A(c->stc_g1, 32 * H); A(c->stc_g2, 32 * H); A(c->stc_tmp, 8 * H * H);
    }
    A(c->embed, (size_t)c->V * H); A(c->final_norm, H); A(c->wheads, (size_t)c->NVpad * H);
    A(c->time_tab, (size_t)c->Tv * H); A(c->score_tab, (size_t)c->Sv * H); A(c->sync_row, H);
    c->llm.resize(c->NL);
    for (auto& l : c->llm) {
        A(l.rms1, H); A(l.wqkv, (size_t)c->QKV * H); A(l.wo, H * H); A(l.rms2, H); A(l.wgu, 2 * I * H); A(l.wd, H * I);
        A(l.wqkv_d, (size_t)c->QKV * H); A(l.wo_d, H * H); A(l.wgu_d, 2 * I * H); A(l.wd_d, H * I);
        if (c->fp8) {
            A(l.wqkv8, (size_t)c->QKV * H); A(l.wo8, H * H); A(l.wgu8, 2 * I * H); A(l.wd8, H * I);
            A(l.wqkv8_d, (size_t)c->QKV * H); A(l.wo8_d, H * H); A(l.wgu8_d, 2 * I * H); A(l.wd8_d, H * I);
            A(l.sqkv, (size_t)c->QKV); A(l.so, H); A(l.sgu, 2 * I); A(l.sd, H);
        }
    }
    A(c->slot_cos, (size_t)c->GG * vh / 2); A(c->slot_sin, (size_t)c->GG * vh / 2);
    A(c->rope_cos, (size_t)c->max_ctx * c->HD / 2); A(c->rope_sin, (size_t)c->max_ctx * c->HD / 2);
    // --- KV cache ---
    c->kv_head_stride = (size_t)c->ctx_pad * c->HD;
    c->slot_stride = c->kv_head_stride * c->NKV;
    c->layer_stride = c->slot_stride * c->max_B;
    A(c->kcache, c->layer_stride * c->NL); A(c->vcache, c->layer_stride * c->NL);
    // --- ViT workspaces ---
    const size_t Tm = cfg->max_frames, Tv_ = c->vit_frames, Mv = Tv_ * c->NT;      // tower workspaces: vit_batch_frames at a time
    A(c->vX, Mv * vh); A(c->vH, Mv * vh); A(c->vQKV, Mv * 3 * vh); A(c->vVT, Tv_ * vh * c->tokpad);
    {
        const size_t mlp = Mv * vi, im2 = Tv_ * c->GG * c->Kpad;
        A(c->vMLP, mlp > im2 ? mlp : im2);
    }
    A(c->sl_res, Tm * c->S * vh); A(c->sl_out, Tm * c->S * H);
    if (!c->stc) { c->sl_ws_floats = launch_slot_pool_ws_floats((int)Tm, (int)vh); A(c->sl_ws, c->sl_ws_floats); }
    {
        const int g2 = c->G / 2 + 1;
        const size_t vr = c->stc ? (size_t)(Tm / 2 + 1) * g2 * g2 : Tm * c->TPF;
        A(c->video, vr * H);
    }
    // --- prefill workspaces ---
    const size_t Lm = c->max_ctx;
    // prefill workspaces hold PF_MAX sequences (trace_llm_prefill_pair / _multi)
    const size_t PR = pf_rows(c->max_ctx);
    A(c->pX, PR * H); A(c->pH, PR * H); A(c->pQKV, PR * c->QKV);
    A(c->pO, PR * H); A(c->pACT, PR * I);
    A(c->d_kind, Lm); A(c->d_row, Lm);
    if (c->fp8) { A(c->pA8, PR * std::max(H, I)); A(c->psa, PR); A(c->dA8, (size_t)SK_ROWS * std::max(H, I)); A(c->dsa, SK_ROWS); A(c->dH8, (size_t)SK_ROWS * H); A(c->dsh, SK_ROWS); }
    // --- decode ---
    A(c->dX, SK_ROWS * H); A(c->dH, SK_ROWS * H); A(c->dQKV, SK_ROWS * (size_t)c->QKV); A(c->dO, SK_ROWS * H); A(c->dACT, SK_ROWS * I);
    A(c->xlast, (size_t)std::max(c->max_B, 64) * H);
    A(c->attn_ws, (size_t)SK_ROWS * c->NQ * c->nsplit * (c->HD + 2)); A(c->tickets, SK_ROWS * c->NKV);
    {
        size_t f = skinny_ws_floats(c->QKV, H, EPI_NONE);
        f = std::max(f, skinny_ws_floats(c->QKV, H, EPI_PARTIAL));
        f = std::max(f, skinny_ws_floats(H, H, EPI_PARTIAL));
        f = std::max(f, skinny_ws_floats(2 * I, H, EPI_SWIGLU));
        f = std::max(f, skinny_ws_floats(2 * I, H, EPI_PARTIAL));
        f = std::max(f, skinny_ws_floats(H, I, EPI_PARTIAL));
        if (c->fp8)      // the fp8 GEMV picks its own K-chunk count (128-k units): size the partial rows for it as well
            for (int B : {1, 16, 17, 32, 33, 64}) {
                f = std::max(f, (size_t)skinny_w8_ks(c->QKV, (int)H, B) * SK_ROWS * c->QKV);
                f = std::max(f, (size_t)skinny_w8_ks((int)H, (int)H, B) * SK_ROWS * H);
                f = std::max(f, (size_t)skinny_w8_ks(2 * (int)I, (int)H, B) * SK_ROWS * 2 * I);
                f = std::max(f, (size_t)skinny_w8_ks((int)H, (int)I, B) * SK_ROWS * H);
                f = std::max(f, (size_t)skinny_fp8_ks(c->QKV, (int)H, B) * SK_ROWS * c->QKV);
                f = std::max(f, (size_t)skinny_fp8_ks((int)H, (int)H, B) * SK_ROWS * H);
                f = std::max(f, (size_t)skinny_fp8_ks(2 * (int)I, (int)H, B) * SK_ROWS * 2 * I);
                f = std::max(f, (size_t)skinny_fp8_ks((int)H, (int)I, B) * SK_ROWS * H);
            }
        // batches above SKINNY_ROWS: the split-K partial-row GEMM's chunks (qkv, o, down)
        f = std::max(f, (size_t)gemm_partial_ks(c->QKV, (int)H) * SK_ROWS * c->QKV);
        f = std::max(f, (size_t)gemm_partial_ks((int)H, (int)H) * SK_ROWS * H);
        f = std::max(f, (size_t)gemm_partial_ks((int)H, (int)I) * SK_ROWS * H);
        c->sk_ws_floats = std::max<size_t>(f, 64);
        c->sk_ntickets = (int)std::max<size_t>(std::max<size_t>((size_t)c->QKV, (size_t)H), (size_t)(2 * I)) / 16;
        A(c->sk_ws, c->sk_ws_floats); A(c->sk_tickets, c->sk_ntickets);
        A(c->sk_ws2, c->sk_ws_floats); A(c->dX2, SK_ROWS * H);
    }
    c->ntiles = c->NVpad / 16;
    A(c->part_val, (size_t)SK_ROWS * c->ntiles); A(c->part_idx, (size_t)SK_ROWS * c->ntiles);
    // trace_llm_head_logits: its own arg-max partials and three constant head-id rows (all 0 / all 1 / all 2), so that it shares nothing with a
    // decode batch in flight on another stream and needs no host copy or synchronisation per call
    A(c->hl_val, (size_t)SK_ROWS * c->ntiles); A(c->hl_idx, (size_t)SK_ROWS * c->ntiles);
    A(c->d_heads_tmp, 3 * SK_ROWS); A(c->d_slots, SK_ROWS); A(c->d_pos, SK_ROWS); A(c->d_heads, SK_ROWS); A(c->d_done, SK_ROWS); A(c->d_out_len, SK_ROWS); A(c->d_step, 4); A(c->d_params, 4);
    A(c->d_out_ids, (size_t)SK_ROWS * cfg->max_new_tokens); A(c->d_forced, (size_t)SK_ROWS * cfg->max_new_tokens);
#undef A
    if (rc == TRACE_OK && hipHostMalloc((void**)&c->h_ring, Lm * 8 * trace_ctx::NSTAGE_H) != hipSuccess) rc = fail(TRACE_ERR_HIP, "hipHostMalloc");
    if (rc != TRACE_OK) { trace_ctx_destroy(c); return rc; }
    c->h_len = Lm;
    c->h_kind = c->h_ring; c->h_row = c->h_kind + Lm;
    for (int i = 0; i < trace_ctx::NSTAGE_H; ++i) hipEventCreateWithFlags(&c->h_ev[i], hipEventDisableTiming);
    hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking);
    hipEventCreate(&c->ev0); hipEventCreate(&c->ev1);
    hipEventCreate(&c->gev0); hipEventCreate(&c->gev1);
    hipEventCreate(&c->mev0); hipEventCreate(&c->mev1);
    for (auto& e : c->vev) hipEventCreate(&e);
    c->kev.resize(1024);
    for (auto& e : c->kev) hipEventCreate(&e);
    if (device_id >= 0 && device_id < 16) { g_ctx_per_dev[device_id] += 1; c->counted = true; }
    *out = c;
    return TRACE_OK;
}

extern "C" int trace_ctx_destroy(trace_ctx* c) {
    if (!c) return TRACE_OK;
    (void)hipSetDevice(c->dev);       // the calling thread's current device may be another GPU: everything below (and the counters' key) is this context's
    hipDeviceSynchronize();
    for (auto& st : c->streams) { gemm_pers_forget(st); hipStreamDestroy(st); }
    for (auto& g : c->graphs) if (g) hipGraphExecDestroy(g);
    for (auto& e : c->kev) if (e) hipEventDestroy(e);
    for (auto& e : c->vev) if (e) hipEventDestroy(e);
    if (c->mev0) hipEventDestroy(c->mev0);
    if (c->mev1) hipEventDestroy(c->mev1);
    if (c->gev0) hipEventDestroy(c->gev0);
    if (c->gev1) hipEventDestroy(c->gev1);
    for (void* p : c->allocs) hipFree(p);
    if (c->pp_buf) hipFree(c->pp_buf);
    if (c->h_ring) hipHostFree(c->h_ring);
    for (int i = 0; i < trace_ctx::NSTAGE_H; ++i) if (c->h_ev[i]) hipEventDestroy(c->h_ev[i]);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->counted && c->dev >= 0 && c->dev < 16 && --g_ctx_per_dev[c->dev] == 0) gemm_pers_release(c->dev);
    delete c;
    return TRACE_OK;
}

extern "C" int64_t trace_ctx_device_bytes(trace_ctx* c) { return c ? (int64_t)c->total_bytes : 0; }

// ------------------------------------------------------------------------------------------------ weights
static int copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int on_device) {
    HIPCHK(hipMemcpy2D(dst, dpitch, src, spitch, width, height, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return TRACE_OK;
}
static int copy1d(void* dst, const void* src, size_t bytes, int on_device) {
    HIPCHK(hipMemcpy(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return TRACE_OK;
}
static bool starts(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }

extern "C" int trace_ctx_load_tensor(trace_ctx* c, const char* name_, const void* data, int on_device, const int64_t* shape,
                                     int ndim) {
    if (!c || !name_ || !data) return fail(TRACE_ERR_ARG, "null argument");
    std::string name(name_);
    int64_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    const size_t H = c->H, I = c->I, vh = c->vh, vi = c->vi, E = sizeof(bf16_t);
    auto expect = [&](int64_t n) -> int {
        if (numel != n) return fail(TRACE_ERR_ARG, "tensor " + name + ": expected " + std::to_string(n) + " elements, got " + std::to_string(numel));
        return TRACE_OK;
    };
    auto done = [&]() { c->loaded[name] = 1; return TRACE_OK; };
    auto flat = [&](bf16_t* dst, int64_t n) -> int { TRY(expect(n)); TRY(copy1d(dst, data, n * E, on_device)); return done(); };

    if (name == "model.embed_tokens.weight") return flat(c->embed, (int64_t)c->V * H);
    if (name == "model.norm.weight") return flat(c->final_norm, H);
    if (name == "lm_head.weight") return flat(c->wheads, (int64_t)c->V * H);
    if (name == "sync_head.weight") return flat(c->wheads + (size_t)c->V * H, H);
    if (name == "time_head.weight") return flat(c->wheads + (size_t)(c->V + 1) * H, (int64_t)c->Tv * H);
    if (name == "score_head.weight") return flat(c->wheads + (size_t)(c->V + 1 + c->Tv) * H, (int64_t)c->Sv * H);
    if (name == "model.time_tower.embed_tokens.weight") return flat(c->time_tab, (int64_t)c->Tv * H);
    if (name == "model.score_tower.embed_tokens.weight") return flat(c->score_tab, (int64_t)c->Sv * H);
    if (name == "model.sync_tower.embed_tokens.weight") return flat(c->sync_row, H);
    if (c->stc && starts(name, "model.mm_projector.")) {
        const std::string k = name.substr(strlen("model.mm_projector."));
        if (k == "sampler.0.bias") return flat(c->stc_b3d, H);
        if (k == "readout.0.weight") return flat(c->stc_r0w, H * H);
        if (k == "readout.0.bias") return flat(c->stc_r0b, H);
        if (k == "readout.2.weight") return flat(c->stc_r2w, H * H);
        if (k == "readout.2.bias") return flat(c->stc_r2b, H);
        if (k == "sampler.0.weight") {      // [Co, Ci, 2,2,2] -> [Co][tap][Ci] for the im2col GEMM
            TRY(expect(8 * (int64_t)H * H));
            TRY(copy1d(c->stc_tmp, data, 8 * H * H * E, on_device));
            if (launch_permute_conv3d_w(c->stc_tmp, c->stc_w3d, (int)H, (int)H, 0) != TRACE_OK) return fail(TRACE_ERR_HIP, "permute launch");
            HIPCHK(hipDeviceSynchronize());
            return done();
        }
        if (k.size() > 6 && (k[0] == 's') && (k[1] == '1' || k[1] == '2') && k[2] == '.' && k[3] == 'b' && k[5] == '.') {
            const int st = k[1] - '1', b = k[4] - '1';
            if (b < 0 || b > 3) return fail(TRACE_ERR_ARG, "bad STC block in " + name);
            StcBlock& q = c->stc_blk[st][b];
            const std::string t = k.substr(6);
            if (t == "conv1.conv.weight") return flat(q.w1, (int64_t)H * q.cin);
            if (t == "conv1.bn.weight") return flat(q.n1w, H);
            if (t == "conv1.bn.bias") return flat(q.n1b, H);
            if (t == "conv2.conv.weight") return flat(q.wdw, H * 9);
            if (t == "conv2.bn.weight") return flat(q.n2w, H);
            if (t == "conv2.bn.bias") return flat(q.n2b, H);
            if (t == "se.fc1.weight") return flat(q.fc1w, (int64_t)q.rd * H);
            if (t == "se.fc1.bias") return flat(q.fc1b, q.rd);
            if (t == "se.fc2.weight") return flat(q.fc2w, (int64_t)H * q.rd);
            if (t == "se.fc2.bias") return flat(q.fc2b, H);
            if (t == "conv3.conv.weight") return flat(q.w3, H * H);
            if (t == "conv3.bn.weight") return flat(q.n3w, H);
            if (t == "conv3.bn.bias") return flat(q.n3b, H);
            if (q.wd && t == "downsample.conv.weight") return flat(q.wd, (int64_t)H * q.cin);
            if (q.wd && t == "downsample.bn.weight") return flat(q.ndw, H);
            if (q.wd && t == "downsample.bn.bias") return flat(q.ndb, H);
        }
        return fail(TRACE_ERR_ARG, "unknown tensor " + name);
    }
    if (name == "model.mm_projector.slots") return flat(c->sl_slots, (int64_t)vh * c->S);
    if (name == "model.mm_projector.ln_vision.weight") return flat(c->sl_lnw, vh);
    if (name == "model.mm_projector.ln_vision.bias") return flat(c->sl_lnb, vh);
    if (name == "model.mm_projector.readout.weight") return flat(c->sl_readout, (int64_t)H * vh);

    if (starts(name, "model.layers.")) {
        const char* p = name.c_str() + strlen("model.layers.");
        char* end = nullptr;
        const long l = strtol(p, &end, 10);
        if (l < 0 || l >= c->NL || *end != '.') return fail(TRACE_ERR_ARG, "bad layer index in " + name);
        const std::string k(end + 1);
        LlmLayer& L = c->llm[l];
        const size_t qrows = (size_t)c->NQ * c->HD, kvrows = (size_t)c->NKV * c->HD;
        if (k == "input_layernorm.weight") return flat(L.rms1, H);
        if (k == "post_attention_layernorm.weight") return flat(L.rms2, H);
        if (k == "self_attn.q_proj.weight") return flat(L.wqkv, qrows * H);
        if (k == "self_attn.k_proj.weight") return flat(L.wqkv + qrows * H, kvrows * H);
        if (k == "self_attn.v_proj.weight") return flat(L.wqkv + (qrows + kvrows) * H, kvrows * H);
        if (k == "self_attn.o_proj.weight") return flat(L.wo, H * H);
        if (k == "mlp.down_proj.weight") return flat(L.wd, H * I);
        if (k == "mlp.gate_proj.weight" || k == "mlp.up_proj.weight") {
            // interleave 16-row groups: packed rows [32t, 32t+16) = gate[16t..], [32t+16, 32t+32) = up[16t..]
            TRY(expect(I * H));
            bf16_t* dst = L.wgu + (k == "mlp.up_proj.weight" ? 16 * H : 0);
            TRY(copy2d(dst, 32 * H * E, data, 16 * H * E, 16 * H * E, I / 16, on_device));
            return done();
        }
        return fail(TRACE_ERR_ARG, "unknown tensor " + name);
    }

    // vision tower: accept both transformers key layouts (with / without ".vision_model")
    const char* vp1 = "model.vision_tower.vision_tower.vision_model.";
    const char* vp2 = "model.vision_tower.vision_tower.";
    std::string k;
    if (starts(name, vp1)) k = name.substr(strlen(vp1));
    else if (starts(name, vp2)) k = name.substr(strlen(vp2));
    else return fail(TRACE_ERR_ARG, "unknown tensor " + name);
    auto vdone = [&](const std::string& canon) { c->loaded[std::string(vp1) + canon] = 1; return TRACE_OK; };
    auto vflat = [&](bf16_t* dst, int64_t n) -> int { TRY(expect(n)); TRY(copy1d(dst, data, n * E, on_device)); return vdone(k); };
    if (k == "embeddings.class_embedding") return vflat(c->cls, vh);
    if (k == "embeddings.position_embedding.weight") return vflat(c->pos_emb, (int64_t)c->NT * vh);
    if (k == "embeddings.patch_embedding.weight") {
        TRY(expect((int64_t)vh * c->Kpatch));
        TRY(copy2d(c->patch_w, c->Kpad * E, data, c->Kpatch * E, c->Kpatch * E, vh, on_device));
        return vdone(k);
    }
    if (k == "pre_layrnorm.weight") return vflat(c->pre_w, vh);
    if (k == "pre_layrnorm.bias") return vflat(c->pre_b, vh);
    if (starts(k, "post_layernorm.") || k == "embeddings.position_ids") return 1;       // unused on this path
    if (starts(k, "encoder.layers.")) {
        const char* p = k.c_str() + strlen("encoder.layers.");
        char* end = nullptr;
        const long l = strtol(p, &end, 10);
        if (l < 0 || *end != '.') return fail(TRACE_ERR_ARG, "bad layer index in " + name);
        if (l >= c->vL) return 1;     // layers after the selected hidden state never run (select_layer = -2)
        const std::string kk(end + 1);
        VitLayer& L = c->vit[l];
        if (kk == "layer_norm1.weight") return vflat(L.ln1w, vh);
        if (kk == "layer_norm1.bias") return vflat(L.ln1b, vh);
        if (kk == "layer_norm2.weight") return vflat(L.ln2w, vh);
        if (kk == "layer_norm2.bias") return vflat(L.ln2b, vh);
        if (kk == "self_attn.q_proj.weight") return vflat(L.wqkv, vh * vh);
        if (kk == "self_attn.k_proj.weight") return vflat(L.wqkv + vh * vh, vh * vh);
        if (kk == "self_attn.v_proj.weight") return vflat(L.wqkv + 2 * vh * vh, vh * vh);
        if (kk == "self_attn.q_proj.bias") return vflat(L.bqkv, vh);
        if (kk == "self_attn.k_proj.bias") return vflat(L.bqkv + vh, vh);
        if (kk == "self_attn.v_proj.bias") return vflat(L.bqkv + 2 * vh, vh);
        if (kk == "self_attn.out_proj.weight") return vflat(L.wo, vh * vh);
        if (kk == "self_attn.out_proj.bias") return vflat(L.bo, vh);
        if (kk == "mlp.fc1.weight") return vflat(L.w1, vi * vh);
        if (kk == "mlp.fc1.bias") return vflat(L.b1, vi);
        if (kk == "mlp.fc2.weight") return vflat(L.w2, vh * vi);
        if (kk == "mlp.fc2.bias") return vflat(L.b2, vh);
    }
    return fail(TRACE_ERR_ARG, "unknown tensor " + name);
}

extern "C" int trace_ctx_finalize(trace_ctx* c) {
    if (!c) return fail(TRACE_ERR_ARG, "null ctx");
    const int expected = (c->stc ? 9 + 8 * 13 + 3 + 6 : 13) + 9 * c->NL + 5 + 16 * c->vL;
    if ((int)c->loaded.size() != expected)
        return fail(TRACE_ERR_STATE, "weights incomplete: " + std::to_string(c->loaded.size()) + " of " + std::to_string(expected) + " tensors loaded");
    // RoPE tables, computed the way the reference does (fp32 inv_freq, fp32 angle, cos/sin of that angle)
    {
        const int d = c->vh, h2 = d / 2, n = c->GG;
        std::vector<float> cs((size_t)n * h2), sn((size_t)n * h2);
        for (int j = 0; j < h2; ++j) {
            const float inv = 1.0f / (float)pow((double)c->c.slot_rope_base, (double)((float)(2 * j) / (float)d));
            for (int t = 0; t < n; ++t) {
                const float ang = (float)t * inv;
                cs[(size_t)t * h2 + j] = (float)cos((double)ang);
                sn[(size_t)t * h2 + j] = (float)sin((double)ang);
            }
        }
        HIPCHK(hipMemcpy(c->slot_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->slot_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    }
    {
        const int d = c->HD, h2 = d / 2, n = c->max_ctx;
        std::vector<float> cs((size_t)n * h2), sn((size_t)n * h2);
        for (int j = 0; j < h2; ++j) {
            const float inv = 1.0f / (float)pow((double)c->c.rope_theta, (double)((float)(2 * j) / (float)d));
            for (int t = 0; t < n; ++t) {
                const float ang = (float)t * inv;
                cs[(size_t)t * h2 + j] = (float)cos((double)ang);
                sn[(size_t)t * h2 + j] = (float)sin((double)ang);
            }
        }
        HIPCHK(hipMemcpy(c->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    }
    // fused patch embedding: the conv weight in the kernel's k order / fragment layout, and the CLS row (the same for every frame)
    if (c->patch_wp) {
        LCHK(launch_patch_pack(c->patch_w, c->Kpad, c->patch_wp, c->vh, c->P, 0));
        LCHK(launch_cls_row(c->cls, c->pos_emb, c->pre_w, c->pre_b, c->cls_row, c->vh, c->c.v_eps, 0));
    }
    // decode copies of the LLM matrices in the GEMV tile layout (288 GB of HBM: +14.5 GB buys ~25% on the weight stream)
    for (auto& l : c->llm) {
        LCHK(launch_tile_pack(l.wqkv, c->H, l.wqkv_d, c->QKV, c->H, 0));
        LCHK(launch_tile_pack(l.wo, c->H, l.wo_d, c->H, c->H, 0));
        LCHK(launch_tile_pack(l.wgu, c->H, l.wgu_d, 2 * c->I, c->H, 0));
        LCHK(launch_tile_pack(l.wd, c->I, l.wd_d, c->H, c->I, 0));
        if (c->fp8) {      // per-output-row e4m3 quantisation of the four projections (from the bf16 copies), then the decode tile layout
            LCHK(launch_quant_rows_fp8(l.wqkv, c->H, l.wqkv8, c->H, l.sqkv, c->QKV, c->H, 0));
            LCHK(launch_quant_rows_fp8(l.wo, c->H, l.wo8, c->H, l.so, c->H, c->H, 0));
            LCHK(launch_quant_rows_fp8(l.wgu, c->H, l.wgu8, c->H, l.sgu, 2 * c->I, c->H, 0));
            LCHK(launch_quant_rows_fp8(l.wd, c->I, l.wd8, c->I, l.sd, c->H, c->I, 0));
            LCHK(launch_tile_pack_fp8(l.wqkv8, c->H, l.wqkv8_d, c->QKV, c->H, 0));
            LCHK(launch_tile_pack_fp8(l.wo8, c->H, l.wo8_d, c->H, c->H, 0));
            LCHK(launch_tile_pack_fp8(l.wgu8, c->H, l.wgu8_d, 2 * c->I, c->H, 0));
            LCHK(launch_tile_pack_fp8(l.wd8, c->I, l.wd8_d, c->H, c->I, 0));
        }
    }
    {
        int32_t h[3 * SK_ROWS];
        for (int i = 0; i < 3 * SK_ROWS; ++i) h[i] = i / SK_ROWS;
        HIPCHK(hipMemcpy(c->d_heads_tmp, h, sizeof(h), hipMemcpyHostToDevice));
    }
    // ticket counters of the persistent GEMM for the streams this context launches on by itself (an allocation + a memset: not something to
    // meet inside a timed or captured region); a caller's own stream gets its counters at its first launch
    HIPCHK(hipSetDevice(c->dev));
    if (gemm_pers_init(nullptr) != TRACE_OK || gemm_pers_init(c->cap_stream) != TRACE_OK) return fail(TRACE_ERR_HIP, "persistent GEMM ticket counters");
    HIPCHK(hipDeviceSynchronize());
    c->finalized = true;
    return TRACE_OK;
}

// ------------------------------------------------------------------------------------------------ ViT
int g_vit_patch_fused = 1;  // 0: im2col matrix -> GEMM -> assemble instead of the fused front end (A/B: trace_op_set_gemm_variant(160 + x))
static unsigned long long* g_gemm_trace = nullptr;      // tools/gemm_trace.py
extern "C" int trace_op_set_gemm_trace(void* buf) { g_gemm_trace = (unsigned long long*)buf; return TRACE_OK; }
static int gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, bf16_t* C, int ldc, const bf16_t* bias, const bf16_t* R,
                int ldr, int M, int N, int K, int epi, hipStream_t s) {
    GemmArgs g{A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, g_gemm_trace, 0, nullptr, nullptr};
    const int rc = launch_gemm_bf16(g, epi, s);
    if (rc != TRACE_OK) return fail(rc, "gemm launch failed (M=" + std::to_string(M) + " N=" + std::to_string(N) + " K=" + std::to_string(K) + ")");
    return TRACE_OK;
}

// C = (quantise_rows(A) . W8^T) * scales (+ epilogue): A [M,K] bf16 is quantised row-wise into c->pA8 / c->psa first
static int gemm_fp8(trace_ctx* c, const bf16_t* A, int lda, const uint8_t* W8, const float* sw, bf16_t* C, int ldc, const bf16_t* R, int ldr,
                    int M, int N, int K, int epi, hipStream_t s) {
    LCHK(launch_quant_rows_fp8(A, lda, c->pA8, K, c->psa, M, K, s));
    GemmArgs g{reinterpret_cast<const bf16_t*>(c->pA8), K, reinterpret_cast<const bf16_t*>(W8), K, C, ldc, nullptr, R, ldr, M, N, K, nullptr, 1, c->psa, sw};
    const int rc = launch_gemm_bf16(g, epi, s);
    if (rc != TRACE_OK) return fail(rc, "fp8 gemm launch failed (M=" + std::to_string(M) + " N=" + std::to_string(N) + " K=" + std::to_string(K) + ")");
    return TRACE_OK;
}

extern "C" int trace_vit_forward(trace_ctx* c, const void* frames, int frames_dtype, int T, void* feats_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (!frames || T < 1 || T > c->vit_frames) return fail(TRACE_ERR_ARG, "bad frames / T (more than max_frames / vit_batch_frames)");
    hipStream_t s = (hipStream_t)stream;
    const int vh = c->vh, vi = c->vi, NT = c->NT, GG = c->GG, Mv = T * NT;
    // front end: one kernel reads the frame tensor, multiplies the patches on the MFMA, adds CLS / position embeddings and applies
    // pre_layrnorm (patch_embed.hip).  g_vit_patch_fused = 0 (A/B) or a patch size / width the kernel does not take: the round-1 path,
    // im2col matrix -> GEMM -> assemble.
    const bool fused_pe = g_vit_patch_fused && c->patch_wp;
    if (fused_pe) {
        LCHK(launch_patch_embed(frames, frames_dtype == 1, c->patch_wp, c->pos_emb, c->pre_w, c->pre_b, c->cls_row, c->vX, T, c->c.v_image, c->P, vh,
                                c->c.v_eps, s));
    } else {
        bf16_t* im2 = c->vMLP;                       // [T*GG, Kpad]
        bf16_t* pe = c->vH;                          // [T*GG, vh]
        LCHK(launch_im2col(frames, frames_dtype == 1, im2, T, c->c.v_image, c->P, c->Kpad, s));
        TRY(gemm(im2, c->Kpad, c->patch_w, c->Kpad, pe, vh, nullptr, nullptr, 0, T * GG, vh, c->Kpad, EPI_NONE, s));
        LCHK(launch_vit_assemble(pe, c->cls, c->pos_emb, c->pre_w, c->pre_b, c->vX, T, GG, vh, c->c.v_eps, s));
    }
    AttnArgs a{};
    a.Q = c->vQKV; a.K = c->vQKV + vh; a.V = c->vVT; a.O = c->vH;
    a.q_bs = (long)NT * 3 * vh; a.q_hs = 64; a.q_rs = 3 * vh;
    a.k_bs = a.q_bs; a.k_hs = 64; a.k_rs = 3 * vh;
    a.v_bs = (long)vh * c->tokpad; a.v_hs = 64L * c->tokpad; a.v_rs = c->tokpad;
    a.o_bs = (long)NT * vh; a.o_hs = 64; a.o_rs = vh;
    a.nq_rows = NT; a.nkv_rows = NT; a.batch = T; a.heads = c->vheads; a.kv_heads = c->vheads;
    a.scale = 0.125f; a.causal = 0;
    a.Vrow = c->vQKV + 2 * vh; a.vr_bs = a.q_bs; a.vr_hs = 64; a.vr_rs = 3 * vh;
    a.v_perm = attn_vit_wants_perm(NT, true) ? (attn_vit_rowmajor_v() ? 2 : 1) : 0;     // 2: V read row-major from vQKV, no transpose pass
    for (int l = 0; l < c->vL; ++l) {
        const VitLayer& L = c->vit[l];
        // MFMA roofline probe (profile == 2): HIP events around ONE launch of each of the layer's four GEMM shapes, in layer 0, per call — of the
        // largest call shape seen since trace_set_profile only (the short tail call of a frame stream would mix two shapes into one average).
        // The fc1 pair (mev0 / mev1) is the `roofline` object's bracket; pairs 0..2 of vev are qkv, out-proj and fc2.
        const bool vprof = c->profile == 2 && (c->bracket_mask & 1);
        if (l == 0 && vprof && Mv > c->mM) { c->mM = Mv; c->msum_ms = 0.0; c->msamples = 0; c->vsum_ms[0] = c->vsum_ms[1] = c->vsum_ms[2] = 0.0; }
        const bool probe = (l == 0 && vprof && Mv == c->mM);
#define VPROBE_BEGIN(i) if (probe) hipEventRecord(c->vev[2 * (i)], s)
#define VPROBE_END(i, N_, K_) if (probe) { hipEventRecord(c->vev[2 * (i) + 1], s); c->vflops[i] = 2.0 * Mv * (double)(N_) * (K_); }
        LCHK(launch_layernorm(c->vX, vh, c->vH, vh, L.ln1w, L.ln1b, Mv, vh, c->c.v_eps, s));
        VPROBE_BEGIN(0);
        TRY(gemm(c->vH, vh, L.wqkv, vh, c->vQKV, 3 * vh, L.bqkv, nullptr, 0, Mv, 3 * vh, vh, EPI_NONE, s));
        VPROBE_END(0, 3 * vh, vh);
        if (a.v_perm != 2)
            LCHK(launch_transpose_v(c->vQKV + 2 * vh, (long)NT * 3 * vh, 64, 3 * vh, c->vVT, a.v_bs, a.v_hs, c->tokpad, NT, 64,
                                    c->vheads, T, s, a.v_perm));
        LCHK(launch_attn_vit(a, s));
        VPROBE_BEGIN(1);
        TRY(gemm(c->vH, vh, L.wo, vh, c->vX, vh, L.bo, c->vX, vh, Mv, vh, vh, EPI_RESIDUAL, s));
        VPROBE_END(1, vh, vh);
        LCHK(launch_layernorm(c->vX, vh, c->vH, vh, L.ln2w, L.ln2b, Mv, vh, c->c.v_eps, s));
        if (probe) hipEventRecord(c->mev0, s);
        TRY(gemm(c->vH, vh, L.w1, vh, c->vMLP, vi, L.b1, nullptr, 0, Mv, vi, vh, EPI_QUICKGELU, s));
        if (probe) { hipEventRecord(c->mev1, s); c->mflops = 2.0 * Mv * (double)vi * vh; }
        VPROBE_BEGIN(2);
        TRY(gemm(c->vMLP, vi, L.w2, vi, c->vX, vh, L.b2, c->vX, vh, Mv, vh, vi, EPI_RESIDUAL, s));
        VPROBE_END(2, vh, vi);
#undef VPROBE_BEGIN
#undef VPROBE_END
    }
    if (c->profile == 2 && (c->bracket_mask & 1) && c->vL > 0 && Mv == c->mM) {
        hipEventSynchronize(c->vev[5]);                       // layer 0's fc2: the last of the bracketed launches
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->mev0, c->mev1) == hipSuccess) { c->msum_ms += ms; c->msamples += 1; }
        for (int i = 0; i < 3; ++i)
            if (hipEventElapsedTime(&ms, c->vev[2 * i], c->vev[2 * i + 1]) == hipSuccess) c->vsum_ms[i] += ms;
        c->prof[5] = (float)(c->msum_ms / c->msamples);
        c->prof[6] = (float)c->msamples;
        c->prof[7] = (float)(c->mflops / 1e9);               // GFLOP of the bracketed launch
        c->prof[9] = 0.f;                                     // (was: LayerNorm fold on / off — the fold left the product in round 5)
        for (int i = 0; i < 3; ++i) { c->prof[12 + i] = (float)(c->vsum_ms[i] / c->msamples); c->prof[15 + i] = (float)(c->vflops[i] / 1e9); }
    }
    if (feats_out)   // drop CLS: [T, GG, vh]
        HIPCHK(hipMemcpy2DAsync(feats_out, (size_t)GG * vh * 2, c->vX + vh, (size_t)NT * vh * 2, (size_t)GG * vh * 2, T,
                                hipMemcpyDeviceToDevice, s));
    return TRACE_OK;
}

extern "C" int trace_slot_pool(trace_ctx* c, const void* feats, int T, void* slots_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (T < 1 || T > c->c.max_frames) return fail(TRACE_ERR_ARG, "bad T");
    if (c->stc) return fail(TRACE_ERR_STATE, "this context holds the STC connector, not SpatialSlotPool");
    hipStream_t s = (hipStream_t)stream;
    const int vh = c->vh;
    const bf16_t* f = feats ? (const bf16_t*)feats : c->vX + vh;
    const long fs = feats ? (long)c->GG * vh : (long)c->NT * vh;
    LCHK(launch_slot_pool(f, fs, vh, c->sl_lnw, c->sl_lnb, c->sl_slots, c->slot_cos, c->slot_sin, c->sl_res, T, c->GG, vh,
                          c->S, c->c.slot_eps, c->sl_ws, c->sl_ws_floats, s));
    TRY(gemm(c->sl_res, vh, c->sl_readout, vh, c->sl_out, c->H, nullptr, nullptr, 0, T * c->S, c->H, vh, EPI_NONE, s));
    if (slots_out) HIPCHK(hipMemcpyAsync(slots_out, c->sl_out, (size_t)T * c->S * c->H * 2, hipMemcpyDeviceToDevice, s));
    return TRACE_OK;
}

// one timm-style RegNet bottleneck on channels-last rows: x [N*HW, cin] -> out [N*HW, H] (out may alias nothing of x)
static int stc_block(trace_ctx* c, const StcBlock& k, const bf16_t* x, bf16_t* out, int N, int hh, int ww, hipStream_t s) {
    const int H = c->H, HW = hh * ww, rows = N * HW;
    bf16_t *y = c->stc_y, *z = c->stc_z, *sc = c->stc_sc;
    TRY(gemm(x, k.cin, k.w1, k.cin, y, H, nullptr, nullptr, 0, rows, H, k.cin, EPI_NONE, s));
    LCHK(launch_layernorm(y, H, y, H, k.n1w, k.n1b, rows, H, 1e-6f, s, 1));
    LCHK(launch_dwconv3x3(y, k.wdw, z, N, hh, ww, H, s));
    LCHK(launch_layernorm(z, H, z, H, k.n2w, k.n2b, rows, H, 1e-6f, s, 1));
    LCHK(launch_avgpool(z, c->stc_pool, N, HW, H, s));
    LCHK(launch_skinny_gemm(c->stc_pool, H, k.fc1w, H, c->stc_g1, k.rd, nullptr, 0, N, k.rd, H, EPI_NONE, 0, SKWS(c), s));
    LCHK(launch_bias_act(c->stc_g1, k.fc1b, N, k.rd, ACT_SILU, s));
    LCHK(launch_skinny_gemm(c->stc_g1, k.rd, k.fc2w, k.rd, c->stc_g2, H, nullptr, 0, N, H, k.rd, EPI_NONE, 0, SKWS(c), s));
    LCHK(launch_bias_act(c->stc_g2, k.fc2b, N, H, ACT_SIGMOID, s));
    LCHK(launch_scale_rows(z, c->stc_g2, N, HW, H, s));
    TRY(gemm(z, H, k.w3, H, y, H, nullptr, nullptr, 0, rows, H, H, EPI_NONE, s));
    const bf16_t* shortcut = x;
    if (k.wd) {
        TRY(gemm(x, k.cin, k.wd, k.cin, sc, H, nullptr, nullptr, 0, rows, H, k.cin, EPI_NONE, s));
        LCHK(launch_layernorm(sc, H, sc, H, k.ndw, k.ndb, rows, H, 1e-6f, s, 0));
        shortcut = sc;
    }
    // out = SiLU(LN(conv3 out) + shortcut) in ONE pass, written where the next block reads it (round 5: was LayerNorm, add + SiLU and a copy —
    // three passes over [rows, H]; same rounding points).  x is dead from here on (out may be x: row-wise in place).
    LCHK(launch_layernorm(y, H, out, H, k.n3w, k.n3b, rows, H, 1e-6f, s, 1, shortcut, H));
    return TRACE_OK;
}

extern "C" int trace_stc_connector(trace_ctx* c, const void* feats, int T, void* out, int* rows_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (!c->stc) return fail(TRACE_ERR_STATE, "context was not created with projector_type = stc_connector");
    if (T < 1 || T > c->c.max_frames || T > 32) return fail(TRACE_ERR_ARG, "bad T");
    hipStream_t s = (hipStream_t)stream;
    const int H = c->H, vh = c->vh, G = c->G, GG = c->GG;
    // gather the patch rows (drop CLS) into a dense [T*GG, vh] buffer
    if (feats) HIPCHK(hipMemcpyAsync(c->stc_x, feats, (size_t)T * GG * vh * 2, hipMemcpyDeviceToDevice, s));
    else HIPCHK(hipMemcpy2DAsync(c->stc_x, (size_t)GG * vh * 2, c->vX + vh, (size_t)c->NT * vh * 2, (size_t)GG * vh * 2, T,
                                 hipMemcpyDeviceToDevice, s));
    // s1: 4 blocks on [T, G, G]; every block reads stc_x and leaves its result there (block 0 reads it with row
    // pitch vh and writes pitch H: the result is produced in stc_y and copied once x is dead)
    for (int b = 0; b < 4; ++b) TRY(stc_block(c, c->stc_blk[0][b], c->stc_x, c->stc_x, T, G, G, s));
    // sampler: Conv3d k = s = 2, p = 1 (+bias) + SiLU as an im2col GEMM
    const int To = T / 2 + 1, Go = G / 2 + 1, rows2 = To * Go * Go;
    LCHK(launch_im2col3d(c->stc_x, c->stc_col, T, G, G, H, To, Go, Go, s));
    TRY(gemm(c->stc_col, 8 * H, c->stc_w3d, 8 * H, c->stc_x, H, c->stc_b3d, nullptr, 0, rows2, H, 8 * H, EPI_NONE, s));
    LCHK(launch_bias_act(c->stc_x, nullptr, rows2, H, ACT_SILU, s));
    for (int b = 0; b < 4; ++b) TRY(stc_block(c, c->stc_blk[1][b], c->stc_x, c->stc_x, To, Go, Go, s));
    // readout MLP
    TRY(gemm(c->stc_x, H, c->stc_r0w, H, c->stc_y, H, c->stc_r0b, nullptr, 0, rows2, H, H, EPI_NONE, s));
    LCHK(launch_bias_act(c->stc_y, nullptr, rows2, H, ACT_GELU, s));
    TRY(gemm(c->stc_y, H, c->stc_r2w, H, c->video, H, c->stc_r2b, nullptr, 0, rows2, H, H, EPI_NONE, s));
    c->video_rows = rows2;
    if (rows_out) *rows_out = rows2;
    if (out) HIPCHK(hipMemcpyAsync(out, c->video, (size_t)rows2 * H * 2, hipMemcpyDeviceToDevice, s));
    return TRACE_OK;
}

// The next pinned staging buffer of the ring (h_kind / h_row point into it afterwards): waits only for the copies that last read THAT buffer, four uses ago —
// normally long done — instead of draining the stream.  stage_release() records the buffer's event behind the copies just queued.
static int stage_acquire(trace_ctx* c) {
    c->h_next = (c->h_next + 1) % trace_ctx::NSTAGE_H;
    HIPCHK(hipEventSynchronize(c->h_ev[c->h_next]));          // (an event never recorded is complete)
    c->h_kind = c->h_ring + (size_t)c->h_next * 2 * c->h_len;
    c->h_row = c->h_kind + c->h_len;
    return TRACE_OK;
}
static int stage_release(trace_ctx* c, hipStream_t s) {
    HIPCHK(hipEventRecord(c->h_ev[c->h_next], s));
    return TRACE_OK;
}

// slot pool on `feats` (nullptr = the tower's internal buffer) + the per-frame time-token rows -> c->video
static int encode_tail(trace_ctx* c, const void* feats, int T, const int32_t* time_ids, void* video_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    hipStream_t s = (hipStream_t)stream;
    TRY(trace_slot_pool(c, feats, T, nullptr, stream));
    const int rows = T * c->TPF;
    TRY(stage_acquire(c));
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < c->TPF; ++j) {
            const int r = t * c->TPF + j;
            if (j < c->S) { c->h_kind[r] = 0; c->h_row[r] = t * c->S + j; }
            else {
                const int id = time_ids[t * 6 + (j - c->S)];
                if (id < 0 || id >= c->Tv) return fail(TRACE_ERR_ARG, "time id out of range");
                c->h_kind[r] = 1; c->h_row[r] = id;
            }
        }
    HIPCHK(hipMemcpyAsync(c->d_kind, c->h_kind, rows * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_row, c->h_row, rows * 4, hipMemcpyHostToDevice, s));
    TRY(stage_release(c, s));
    GatherTabs tabs{};
    tabs.t[0] = c->sl_out; tabs.t[1] = c->time_tab;
    LCHK(launch_gather_rows(tabs, c->d_kind, c->d_row, c->video, rows, c->H, s));
    c->video_rows = rows;
    if (video_out) HIPCHK(hipMemcpyAsync(video_out, c->video, (size_t)rows * c->H * 2, hipMemcpyDeviceToDevice, s));
    return TRACE_OK;
}

extern "C" int trace_encode_video(trace_ctx* c, const void* frames, int frames_dtype, int T, const int32_t* time_ids,
                                  void* video_out, void* stream) {
    if (!time_ids) return fail(TRACE_ERR_ARG, "null time_ids");
    if (c && T > c->c.max_frames) return fail(TRACE_ERR_ARG, "bad T (more frames than max_frames)");
    TRY(trace_vit_forward(c, frames, frames_dtype, T, nullptr, stream));
    return encode_tail(c, nullptr, T, time_ids, video_out, stream);
}

extern "C" int trace_encode_features(trace_ctx* c, const void* feats, int T, const int32_t* time_ids, void* video_out, void* stream) {
    if (!time_ids || !feats) return fail(TRACE_ERR_ARG, "null feats / time_ids");
    return encode_tail(c, feats, T, time_ids, video_out, stream);
}

extern "C" int trace_splice_embeds(trace_ctx* c, const int32_t* ids, int n_ids, const int32_t* time_rows, int n_time,
                                   const int32_t* score_rows, int n_score, int* L_out, void* embeds_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (!ids || n_ids < 1) return fail(TRACE_ERR_ARG, "bad ids");
    hipStream_t s = (hipStream_t)stream;
    int nvid = 0;
    for (int i = 0; i < n_ids; ++i) nvid += (ids[i] == -201 || ids[i] == -200);
    if (nvid != 1) return fail(TRACE_ERR_ARG, "only have one video inputs!");          // trace_arch.py:411
    if (c->video_rows <= 0) return fail(TRACE_ERR_STATE, "no encoded video");
    const int L = n_ids - 1 + c->video_rows;
    if (L > c->max_ctx) return fail(TRACE_ERR_ARG, "spliced prompt longer than max_ctx");
    TRY(stage_acquire(c));
    int r = 0, ti = 0, si = 0;
    for (int i = 0; i < n_ids; ++i) {
        const int id = ids[i];
        if (id == -201 || id == -200) {
            for (int j = 0; j < c->video_rows; ++j, ++r) { c->h_kind[r] = 1; c->h_row[r] = j; }
        } else if (id == -205) { c->h_kind[r] = 4; c->h_row[r] = 0; ++r; }
        else if (id == -203) {
            if (ti >= n_time || !time_rows) return fail(TRACE_ERR_ARG, "more <time> placeholders than time tokens");
            if (time_rows[ti] < 0 || time_rows[ti] >= c->Tv) return fail(TRACE_ERR_ARG, "time token id out of range");     // nn.Embedding raises
            c->h_kind[r] = 2; c->h_row[r] = time_rows[ti++]; ++r;
        } else if (id == -204) {
            if (si >= n_score || !score_rows) return fail(TRACE_ERR_ARG, "more <score> placeholders than score tokens");
            if (score_rows[si] < 0 || score_rows[si] >= c->Sv) return fail(TRACE_ERR_ARG, "score token id out of range");
            c->h_kind[r] = 3; c->h_row[r] = score_rows[si++]; ++r;
        } else {
            const int t = id < 0 ? 0 : id;      // torch.clamp(ids, min=0) (trace_arch.py:417)
            if (t >= c->V) return fail(TRACE_ERR_ARG, "token id out of range");
            c->h_kind[r] = 0; c->h_row[r] = t; ++r;
        }
    }
    // the masked assignment of trace_arch.py:426-427 fails with a shape mismatch unless every supplied token has its placeholder
    if (ti != n_time || si != n_score) return fail(TRACE_ERR_ARG, "fewer <time>/<score> placeholders than supplied tokens");
    HIPCHK(hipMemcpyAsync(c->d_kind, c->h_kind, L * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_row, c->h_row, L * 4, hipMemcpyHostToDevice, s));
    TRY(stage_release(c, s));
    GatherTabs tabs{};
    tabs.t[0] = c->embed; tabs.t[1] = c->video; tabs.t[2] = c->time_tab; tabs.t[3] = c->score_tab; tabs.t[4] = c->sync_row;
    LCHK(launch_gather_rows(tabs, c->d_kind, c->d_row, c->pX, L, c->H, s));
    c->spliced_len = L;
    if (L_out) *L_out = L;
    if (embeds_out) HIPCHK(hipMemcpyAsync(embeds_out, c->pX, (size_t)L * c->H * 2, hipMemcpyDeviceToDevice, s));
    return TRACE_OK;
}

// ------------------------------------------------------------------------------------------------ frame preprocessing
// Pillow's Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BICUBIC filter, in its operation order (double).
#pragma clang fp contract(off)
static double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static int pil_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk) {
    double scale, filterscale;
    scale = filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) { w[x] = pil_bicubic((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double k = ww != 0.0 ? w[x] / ww : w[x];
            kk[(size_t)xx * ksize + x] = k < 0 ? (int)(-0.5 + k * (1 << 22)) : (int)(0.5 + k * (1 << 22));
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

extern "C" int trace_preprocess_frames(trace_ctx* c, const void* frames_u8, int T, int H, int W, int pad_to_square,
                                       const float* image_mean, const float* image_std, void* out, int out_dtype, void* stream) {
    if (!c) return fail(TRACE_ERR_ARG, "null ctx");
    if (!frames_u8 || !out || !image_mean || !image_std || T < 1 || H < 1 || W < 1 || (out_dtype != 0 && out_dtype != 1))
        return fail(TRACE_ERR_ARG, "bad preprocess arguments");
    hipStream_t s = (hipStream_t)stream;
    const int S = c->c.v_image;
    // geometry: expand2square (mm_utils.py:259-270), shortest edge -> S (HF get_resize_output_image_size), centre crop
    int PH = H, PW = W, y0 = 0, x0 = 0;
    if (pad_to_square && H != W) {
        PH = PW = H > W ? H : W;
        if (W > H) y0 = (W - H) / 2; else x0 = (H - W) / 2;
    }
    int nh, nw;
    if (PW <= PH) { nw = S; nh = (int)((double)((long)S * PH) / (double)PW); }
    else { nh = S; nw = (int)((double)((long)S * PW) / (double)PH); }
    const int top = (nh - S) / 2, left = (nw - S) / 2;
    std::vector<int32_t> bh, kh, bv, kv;
    const int ksh = pil_coeffs(PW, nw, bh, kh), ksv = pil_coeffs(PH, nh, bv, kv);
    const int row_first = bv[2 * top], row_last = bv[2 * (top + S - 1)] + bv[2 * (top + S - 1) + 1];
    const int nrows = row_last - row_first;
    // device tables: cropped columns / rows only; vertical bounds relative to the first staged row
    std::vector<int32_t> tab((size_t)S * 2 * 2 + (size_t)S * (ksh + ksv));
    int32_t* tbh = tab.data(); int32_t* tbv = tbh + 2 * S; int32_t* tkh = tbv + 2 * S; int32_t* tkv = tkh + (size_t)S * ksh;
    for (int j = 0; j < S; ++j) {
        tbh[2 * j] = bh[2 * (left + j)]; tbh[2 * j + 1] = bh[2 * (left + j) + 1];
        tbv[2 * j] = bv[2 * (top + j)] - row_first; tbv[2 * j + 1] = bv[2 * (top + j) + 1];
        memcpy(tkh + (size_t)j * ksh, kh.data() + (size_t)(left + j) * ksh, (size_t)ksh * 4);
        memcpy(tkv + (size_t)j * ksv, kv.data() + (size_t)(top + j) * ksv, (size_t)ksv * 4);
    }
    // rescale + normalise of an 8-bit value, in the reference's arithmetic: float32(float64(u) * (1/255)), (x - mean) / std
    std::vector<float> lut(3 * 256);
    uint32_t bg = 0;
    for (int ch = 0; ch < 3; ++ch) {
        for (int u = 0; u < 256; ++u) {
            const float x = (float)((double)u * (1.0 / 255));
            lut[ch * 256 + u] = (x - image_mean[ch]) / image_std[ch];
        }
        bg |= (uint32_t)((int)(image_mean[ch] * 255) & 255) << (8 * ch);       // int(x*255), mm_utils.py:457
    }
    const size_t tab_bytes = tab.size() * 4, lut_bytes = lut.size() * 4, tmp_bytes = (size_t)T * nrows * S * 3;
    if (c->pp_bytes < tab_bytes + lut_bytes + tmp_bytes + 512) {
        HIPCHK(hipStreamSynchronize(s));
        if (c->pp_buf) hipFree(c->pp_buf);
        c->pp_bytes = tab_bytes + lut_bytes + tmp_bytes + 512;
        HIPCHK(hipMalloc(&c->pp_buf, c->pp_bytes));
    }
    char* base = (char*)c->pp_buf;
    int32_t* d_tab = (int32_t*)base;
    float* d_lut = (float*)(base + ((tab_bytes + 255) & ~(size_t)255));
    uint8_t* d_tmp = (uint8_t*)d_lut + ((lut_bytes + 255) & ~(size_t)255);
    HIPCHK(hipMemcpyAsync(d_tab, tab.data(), tab_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_lut, lut.data(), lut_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                      // host vectors go out of scope below
    LCHK(launch_resize_h((const uint8_t*)frames_u8, T, H, W, y0, x0, bg, d_tab, d_tab + 4 * S, ksh, row_first, nrows, S, d_tmp, s));
    LCHK(launch_resize_v_norm(d_tmp, T, nrows, S, d_tab + 2 * S, d_tab + 4 * S + (size_t)S * ksh, ksv, d_lut, out, out_dtype, s));
    return TRACE_OK;
}

// ------------------------------------------------------------------------------------------------ LLM prefill
// nb equal-length sequences laid end to end in pX (rows [b*L, (b+1)*L)) -> slots slot0 .. slot0+nb-1.  Two 1967-row
// prompts give the GEMMs M = 3934: 16 row tiles fill the 256x256 tile grid in whole rounds (gate|up 1792 tiles = 7.0
// rounds instead of 896 = 3.5) and o-proj / down-proj reach the 256^2 kernel.
int g_prefill_last_rows = 1;   // 0: the last decoder layer of a prefill runs over all rows like the others (A/B and the bit-identity test: trace_op_set_gemm_variant(750 + x))
static int prefill_impl(trace_ctx* c, int slot0, int nb, int L, void* hidden_out, hipStream_t s) {
    const int H = c->H, I = c->I, HD = c->HD, QKV = c->QKV, M = nb * L;
    AttnArgs a{};
    a.Q = c->pQKV; a.O = c->pO;
    a.q_bs = (long)L * QKV; a.q_hs = HD; a.q_rs = QKV;
    a.k_bs = (long)c->slot_stride; a.k_hs = (long)c->kv_head_stride; a.k_rs = HD;
    a.v_bs = (long)c->slot_stride; a.v_hs = (long)c->kv_head_stride; a.v_rs = c->ctx_pad;      // V^T straight from the cache
    a.o_bs = (long)L * H; a.o_hs = HD; a.o_rs = H;
    a.nq_rows = L; a.nkv_rows = L; a.batch = nb; a.heads = c->NQ; a.kv_heads = c->NKV;
    a.scale = 1.0f / sqrtf((float)HD); a.causal = 1;
    for (int l = 0; l < c->NL; ++l) {
        const LlmLayer& W = c->llm[l];
        bf16_t* kc = c->kcache + (size_t)l * c->layer_stride;
        bf16_t* vc = c->vcache + (size_t)l * c->layer_stride;
        LCHK(launch_rmsnorm(c->pX, H, c->pH, H, W.rms1, M, H, c->c.rms_eps, s));
        // Round 6: the LAST decoder layer of a prefill whose caller does not ask for the hidden rows.  What is consumed afterwards is the layer's K / V
        // rows of every position (the cache) and the final hidden state of each prompt's LAST row only (the reference computes all L rows of
        // everything, trace_mistral.py:190-200, and then uses logits[:, -1]) — so: the k | v slice of the qkv projection over all M rows, and q,
        // attention, o-proj and the MLP for the nb last rows, on the same kernels reading / writing those rows in place (row stride L x width).
        // Same MFMA tile kernels, same K order, same epilogues: bit-identical to the full layer (tests/test_gpu_parity.py), ~3 % of a prefill saved.
        if (g_prefill_last_rows && l == c->NL - 1 && !hidden_out && !c->fp8 && L > 1) {
            const int KV = 2 * c->NKV * HD, QW = c->NQ * HD;
            const size_t last = (size_t)(L - 1);
            TRY(gemm(c->pH, H, W.wqkv + (size_t)QW * H, H, c->pQKV + QW, QKV, nullptr, nullptr, 0, M, KV, H, EPI_NONE, s));
            TRY(gemm(c->pH + last * H, L * H, W.wqkv, H, c->pQKV + last * QKV, L * QKV, nullptr, nullptr, 0, nb, QW, H, EPI_NONE, s));
            LCHK(launch_rope_kv(c->pQKV, QKV, kc, nullptr, (long)c->slot_stride, (long)c->kv_head_stride, nullptr, nullptr, slot0, 0, M,
                                c->NQ, c->NKV, HD, c->rope_cos, c->rope_sin, L, s));      // (also rotates the stale q of the other rows: 10 us, nobody reads them)
            LCHK(launch_transpose_v(c->pQKV + (size_t)(c->NQ + c->NKV) * HD, (long)L * QKV, HD, QKV, vc + (size_t)slot0 * c->slot_stride,
                                    (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, L, HD, c->NKV, nb, s));
            AttnArgs a1 = a;
            a1.Q = c->pQKV + last * QKV; a1.O = c->pO + last * H;
            a1.nq_rows = 1;                                  // one query row per prompt: row L - 1 attends all L keys (causal offset nkv - nq)
            a1.K = kc + (size_t)slot0 * c->slot_stride;
            a1.V = vc + (size_t)slot0 * c->slot_stride;
            LCHK(launch_attn_prefill(a1, s));
            bf16_t* xl = c->pX + last * H;
            bf16_t* hl = c->pH + last * H;
            TRY(gemm(c->pO + last * H, L * H, W.wo, H, xl, L * H, nullptr, xl, L * H, nb, H, H, EPI_RESIDUAL, s));
            LCHK(launch_rmsnorm(xl, L * H, hl, L * H, W.rms2, nb, H, c->c.rms_eps, s));
            TRY(gemm(hl, L * H, W.wgu, H, c->pACT + last * I, L * I, nullptr, nullptr, 0, nb, 2 * I, H, EPI_SWIGLU, s));
            TRY(gemm(c->pACT + last * I, L * I, W.wd, I, xl, L * H, nullptr, xl, L * H, nb, H, I, EPI_RESIDUAL, s));
            continue;
        }
        if (c->fp8) { TRY(gemm_fp8(c, c->pH, H, W.wqkv8, W.sqkv, c->pQKV, QKV, nullptr, 0, M, QKV, H, EPI_NONE, s)); }
        else TRY(gemm(c->pH, H, W.wqkv, H, c->pQKV, QKV, nullptr, nullptr, 0, M, QKV, H, EPI_NONE, s));
        LCHK(launch_rope_kv(c->pQKV, QKV, kc, nullptr, (long)c->slot_stride, (long)c->kv_head_stride, nullptr, nullptr, slot0, 0, M,
                            c->NQ, c->NKV, HD, c->rope_cos, c->rope_sin, L, s));
        // V goes into the cache transposed ([kvh][hd][ctx_pad]; positions L..Lpad-1 are zero-filled, later overwritten)
        LCHK(launch_transpose_v(c->pQKV + (size_t)(c->NQ + c->NKV) * HD, (long)L * QKV, HD, QKV, vc + (size_t)slot0 * c->slot_stride,
                                (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, L, HD, c->NKV, nb, s));
        a.K = kc + (size_t)slot0 * c->slot_stride;
        a.V = vc + (size_t)slot0 * c->slot_stride;
        LCHK(launch_attn_prefill(a, s));
        if (c->fp8) {
            TRY(gemm_fp8(c, c->pO, H, W.wo8, W.so, c->pX, H, c->pX, H, M, H, H, EPI_RESIDUAL, s));
            LCHK(launch_rmsnorm(c->pX, H, c->pH, H, W.rms2, M, H, c->c.rms_eps, s));
            TRY(gemm_fp8(c, c->pH, H, W.wgu8, W.sgu, c->pACT, I, nullptr, 0, M, 2 * I, H, EPI_SWIGLU, s));
            TRY(gemm_fp8(c, c->pACT, I, W.wd8, W.sd, c->pX, H, c->pX, H, M, H, I, EPI_RESIDUAL, s));
            continue;
        }
        TRY(gemm(c->pO, H, W.wo, H, c->pX, H, nullptr, c->pX, H, M, H, H, EPI_RESIDUAL, s));
        LCHK(launch_rmsnorm(c->pX, H, c->pH, H, W.rms2, M, H, c->c.rms_eps, s));
        TRY(gemm(c->pH, H, W.wgu, H, c->pACT, I, nullptr, nullptr, 0, M, 2 * I, H, EPI_SWIGLU, s));
        TRY(gemm(c->pACT, I, W.wd, I, c->pX, H, nullptr, c->pX, H, M, H, I, EPI_RESIDUAL, s));
    }
    if (hidden_out) LCHK(launch_rmsnorm(c->pX, H, (bf16_t*)hidden_out, H, c->final_norm, M, H, c->c.rms_eps, s));
    for (int b = 0; b < nb; ++b) {
        LCHK(launch_rmsnorm(c->pX + ((size_t)b * L + L - 1) * H, H, c->xlast + (size_t)(slot0 + b) * H, H, c->final_norm, 1, H,
                            c->c.rms_eps, s));
        c->slot_len[slot0 + b] = L;
    }
    return TRACE_OK;
}

extern "C" int trace_llm_prefill(trace_ctx* c, int slot, const void* embeds, int L, void* hidden_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (slot < 0 || slot >= c->max_B || L < 1 || L > c->max_ctx) return fail(TRACE_ERR_ARG, "bad slot / L");
    hipStream_t s = (hipStream_t)stream;
    if (embeds) HIPCHK(hipMemcpyAsync(c->pX, embeds, (size_t)L * c->H * 2, hipMemcpyDeviceToDevice, s));
    return prefill_impl(c, slot, 1, L, hidden_out, s);
}

extern "C" int trace_llm_prefill_pair(trace_ctx* c, int slot0, const void* embeds0, const void* embeds1, int L, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (slot0 < 0 || slot0 + 1 >= c->max_B || L < 1 || L > c->max_ctx || !embeds0 || !embeds1) return fail(TRACE_ERR_ARG, "bad slot / L / embeds");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(c->pX, embeds0, (size_t)L * c->H * 2, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->pX + (size_t)L * c->H, embeds1, (size_t)L * c->H * 2, hipMemcpyDeviceToDevice, s));
    return prefill_impl(c, slot0, 2, L, nullptr, s);
}

// n <= 8 prompts of EQUAL spliced length in one pass while n x L <= max(4 max_ctx, min(8192, 8 max_ctx)) rows (the prefill workspaces) -> KV slots slot0 .. slot0 + n - 1.  Four 1967-row prompts give the GEMMs M = 7868 = 31 row tiles:
// qkv 744 tiles = 2.9 rounds of the 256 CUs (a pair: 384 = 1.5 rounds, a quarter of the second round's CUs idle), o / down 496 = 1.94, gate|up 13.6
extern "C" int trace_llm_prefill_multi(trace_ctx* c, int slot0, const void* const* embeds, int n, int L, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (!embeds || n < 1 || n > PF_MAX_N || slot0 < 0 || slot0 + n > c->max_B || L < 1 || L > c->max_ctx) return fail(TRACE_ERR_ARG, "bad slot / n / L / embeds");
    if ((size_t)n * L > pf_rows(c->max_ctx)) return fail(TRACE_ERR_ARG, "n x L exceeds the prefill workspace (max(4 max_ctx, min(8192, 8 max_ctx)) rows)");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        if (!embeds[i]) return fail(TRACE_ERR_ARG, "null embeds");
        HIPCHK(hipMemcpyAsync(c->pX + (size_t)i * L * c->H, embeds[i], (size_t)L * c->H * 2, hipMemcpyDeviceToDevice, s));
    }
    return prefill_impl(c, slot0, n, L, nullptr, s);
}

// masked logits of R final-norm hidden rows under ONE head: what forward() returns for every position of a sequence
// (trace_mistral.py:190-252) — the decode loop itself only ever needs the last row (trace_decode_begin / _steps)
extern "C" int trace_llm_head_logits(trace_ctx* c, const void* hidden, int R, int head, float* logits_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (!hidden || !logits_out || R < 1) return fail(TRACE_ERR_ARG, "bad hidden / logits_out / R");
    if (head < 0 || head > 2) return fail(TRACE_ERR_ARG, "head must be 0, 1 or 2");
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* x = (const bf16_t*)hidden;
    for (int r0 = 0; r0 < R; r0 += SK_ROWS)
        LCHK(launch_head_logits(x + (size_t)r0 * c->H, c->H, c->wheads, c->H, c->d_heads_tmp + head * SK_ROWS, c->V, c->Tv, c->Sv, c->hl_val, c->hl_idx,
                                logits_out + (size_t)r0 * c->NV, std::min(SK_ROWS, R - r0), s));
    return TRACE_OK;
}

// ------------------------------------------------------------------------------------------------ decode
static StepState step_state(trace_ctx* c) {
    StepState st{};
    st.heads = c->d_heads; st.pos = c->d_pos; st.done = c->d_done; st.out_ids = c->d_out_ids; st.out_len = c->d_out_len;
    st.step = c->d_step; st.forced = c->d_forced; st.params = c->d_params;
    return st;
}

static int head_only(trace_ctx* c, const bf16_t* xn, float* logits_out, hipStream_t s) {
    LCHK(launch_head_logits(xn, c->H, c->wheads, c->H, c->d_heads, c->V, c->Tv, c->Sv, c->part_val, c->part_idx, logits_out,
                            c->B, s));
    return TRACE_OK;
}
static int select_only(trace_ctx* c, int advance, hipStream_t s) {
    LCHK(launch_select_next(c->part_val, c->part_idx, step_state(c), c->embed, c->time_tab, c->score_tab, c->sync_row, c->dX,
                            c->H, c->B, c->H, c->V, c->Tv, c->Sv, advance, s));
    return TRACE_OK;
}
static int head_and_select(trace_ctx* c, const bf16_t* xn, int advance, float* logits_out, hipStream_t s) {
    TRY(head_only(c, xn, logits_out, s));
    if (c->host_mode) return TRACE_OK;      // the host picks the token and calls trace_decode_feed
    return select_only(c, advance, s);
}

// decode attention context split: ~256-320 workgroups (8 kv heads x B x nsplit) fill the CUs; more splits only add
// partial-result traffic and ticket latency (measured, ctx 2100: B=1 16 splits 10.4 us, B=4 8 -> 13 us, B=16 2 -> 26 us,
// B=32 1 -> 45 us; B=32 with 16 splits: 79 us)
int g_attn_decode_nsplit = 0;   // A/B: context splits of the decode attention forced to this number (0 = the rule below; trace_op_set_gemm_variant(850 + n))
static int decode_nsplit(int B) { if (g_attn_decode_nsplit > 0) return g_attn_decode_nsplit; const int n = (40 + B / 2) / B; return n < 1 ? 1 : n > 16 ? 16 : n; }

static int head_and_select(trace_ctx* c, const bf16_t* xn, int advance, float* logits_out, hipStream_t s);
int g_decode_wide_min = 32;   // smallest batch that takes the wide (GEMM) decode step — ms per step, GEMV path vs wide step (r03_decode_gemm_ab.txt): 32 rows 5.39 / 5.30, 48 rows
                                 // 7.19 / 6.51, 64 rows 7.88 / 7.11; A/B: trace_op_set_gemm_variant(140 + x): 65 / 33 / 17 / 32
int g_decode_wide_fuse_qkv = 0; // wide decode step: 1 = the attention's fused prologue sums the qkv partial rows, applies RoPE and appends k / v itself (no qkv_finish launch;
                               // same sums and roundings, bit-identical; A/B: trace_op_set_gemm_variant(144 + x))
int g_decode_gemm_tiled = 21;  // wide decode step, GemmArgs::w_tiled: bit 0 = weights from the decode tile copies (0 = row-major prefill copies), bit 2 = 4-stage
                               // K-tile ring, bit 4 = partial rows stored write-through (sc1): the consumer's kernel boundary has no dirty partial bytes to write back
                               // (round 6, same bits: 10.41 -> 10.33 ms per 128-sequence step; bit 1 = nt weight DMA +2.7 %, bit 3 = nt partial stores +0.4 %: off)
                               // (A/B: trace_op_set_gemm_variant(700 + x), 130 + x for the low three bits)

// One decode step for SKINNY_ROWS < B <= SK_ROWS sequences.  A GEMV that parks its activations in LDS cannot hold more than 64 rows x 1024 k, and
// its fp32 partial rows would grow with the row count; above 64 rows the four projections are small-M GEMMs on the MFMA tile kernel instead
// (gemm.hip, 128x128 tiles, weights from the row-major prefill copies): gate|up with the SwiGLU epilogue straight to bf16 — no partial rows,
// no combine kernel — and qkv / o / down cut along K into gemm_partial_ks() chunks whose fp32 partial rows the same consumers as below sum on
// load.  The weights (14 GB per step) are then streamed once per 128 tokens instead of once per 64: bytes per token 0.50 -> 0.39 GB at
// ctx ~2100, where the KV stream (0.27 GB per token) is the larger part.
static int decode_step_wide(trace_ctx* c, float* logits_out, hipStream_t s) {
    const int H = c->H, I = c->I, HD = c->HD, QKV = c->QKV, B = c->B;
    if (c->fp8) return fail(TRACE_ERR_STATE, "the fp8 weight path decodes at most 64 sequences together");
    const int ks_q = gemm_partial_ks(QKV, H), ks_o = gemm_partial_ks(H, H), ks_d = gemm_partial_ks(H, I);
    const int wt = g_decode_gemm_tiled;      // bit 0: weights from the decode tile copies, bit 1: non-temporal weight loads
    auto pgemm = [&](const bf16_t* A, int lda, const bf16_t* Wrow, const bf16_t* Wtile, int ldw, int N, int K, int ks) -> int {
        GemmArgs g{A, lda, (wt & 1) ? Wtile : Wrow, ldw, nullptr, 0, nullptr, nullptr, 0, B, N, K, nullptr, 0, nullptr, nullptr, 0, c->sk_ws, ks, (wt & 1) ? wt : 0};
        if ((size_t)ks * SK_ROWS * N > c->sk_ws_floats) return fail(TRACE_ERR_STATE, "partial-row workspace too small");
        const int rc = launch_gemm_bf16(g, EPI_PARTIAL, s);
        if (rc != TRACE_OK) return fail(rc, "partial GEMM launch failed (N=" + std::to_string(N) + " K=" + std::to_string(K) + ")");
        return TRACE_OK;
    };
    LCHK(launch_rmsnorm(c->dX, H, c->dH, H, c->llm[0].rms1, B, H, c->c.rms_eps, s));
    for (int l = 0; l < c->NL; ++l) {
        const LlmLayer& W = c->llm[l];
        bf16_t* kc = c->kcache + (size_t)l * c->layer_stride;
        bf16_t* vc = c->vcache + (size_t)l * c->layer_stride;
        TRY(pgemm(c->dH, H, W.wqkv, W.wqkv_d, H, QKV, H, ks_q));
        const bool fq = g_decode_wide_fuse_qkv != 0;
        if (!fq) LCHK(launch_qkv_finish(c->sk_ws, ks_q, QKV, c->dQKV, kc, vc, (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, c->d_slots,
                                        c->d_pos, B, c->NQ, c->NKV, c->rope_cos, c->rope_sin, s));
        // roofline probe (profile == 2, eager launches): HIP events around ONE launch of the step's dominant kernel — the layer-0 decode
        // attention, which streams the batch's whole KV cache of that layer
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (l == 0 && c->profile == 2 && (c->bracket_mask & 2) && s != c->cap_stream && c->kev_used + 2 <= (int)c->kev.size()) {
            e0 = c->kev[c->kev_used]; e1 = c->kev[c->kev_used + 1]; c->kev_used += 2;
            c->kbytes_sum += (double)(c->pos_sum + (long)B * (c->step_in_call + 1)) * c->NKV * HD * 2 * 2;   // K + V^T rows of every sequence, bf16
            c->bracket_kind = 3;
        }
        if (e0) hipEventRecord(e0, s);
        LCHK(launch_attn_decode(c->dQKV, QKV, kc, vc, (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, c->d_slots, c->d_pos, c->dO,
                                H, c->attn_ws, c->tickets, B, c->NQ, c->NKV, HD, decode_nsplit(B), 1.0f / sqrtf((float)HD), fq ? 1 : 0,
                                fq ? c->rope_cos : nullptr, fq ? c->rope_sin : nullptr, fq ? c->sk_ws : nullptr, fq ? ks_q : 0, s));
        if (e1) hipEventRecord(e1, s);
        TRY(pgemm(c->dO, H, W.wo, W.wo_d, H, H, H, ks_o));
        LCHK(launch_add_rmsnorm(c->sk_ws, ks_o, c->dX, H, c->dX, H, W.rms2, c->dH, H, B, H, c->c.rms_eps, s));
        if (wt & 1) {
            GemmArgs g{c->dH, H, W.wgu_d, H, c->dACT, I, nullptr, nullptr, 0, B, 2 * I, H, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, wt};
            if (launch_gemm_bf16(g, EPI_SWIGLU, s) != TRACE_OK) return fail(TRACE_ERR_HIP, "gate|up GEMM launch failed");
        } else TRY(gemm(c->dH, H, W.wgu, H, c->dACT, I, nullptr, nullptr, 0, B, 2 * I, H, EPI_SWIGLU, s));
        TRY(pgemm(c->dACT, I, W.wd, W.wd_d, I, H, I, ks_d));
        const bf16_t* nw = l + 1 < c->NL ? c->llm[l + 1].rms1 : c->final_norm;
        LCHK(launch_add_rmsnorm(c->sk_ws, ks_d, c->dX, H, c->dX, H, nw, c->dH, H, B, H, c->c.rms_eps, s));
    }
    return head_and_select(c, c->dH, 1, logits_out, s);
}

int g_decode_fuse_swiglu = 1;      // decode_step_fused: SwiGLU inside the down GEMV (bit-identical to swiglu_combine + GEMV; A/B: trace_op_set_gemm_variant(180 + x))
int g_decode_fuse_norm_rows = 1;   // batches up to this size take decode_step_fused (0 = never; A/B: trace_op_set_gemm_variant(170 + rows)).  Measured, ms per step,
                                   // unfused / fused (profiles/r03_decode_small_ab.txt): batch 1 3.596 / 3.531, batch 2 3.698 / 3.781, batch 4 3.888 / 4.327 — every
                                   // workgroup redoes the row sums, which only a single row repays
// One decode step for 1..4 sequences (the reference drivers' own call shape is 1): the two "sum the partial rows + residual -> new residual, RMSNorm"
// kernels of a layer are folded into the GEMVs that consume their output (decode.hip, SkinnyPro), and (round 4) the SwiGLU combine into the down
// GEMV — 4 launches per layer instead of 7:
//   qkv GEMV [sums the previous layer's down partials + residual, input norm] -> attention (RoPE / append / attention) -> o GEMV
//   -> gate|up GEMV [sums the o partials + residual, post-attention norm] -> down GEMV [sums the gate|up partials, SwiGLU]
// Partial rows alternate between sk_ws (written by o / down, read by the fused GEMVs) and sk_ws2 (written by the fused GEMVs, read by the attention /
// the combine); the residual rows alternate between dX and dX2 (a fused GEMV reads one and writes the other: its workgroups all read the whole row).
static int decode_step_fused(trace_ctx* c, float* logits_out, hipStream_t s) {
    const int H = c->H, I = c->I, HD = c->HD, QKV = c->QKV, B = c->B;
    const int ks_q = skinny_ks(QKV, H, EPI_PARTIAL, B), ks_o = skinny_ks(H, H, EPI_PARTIAL, B);
    const int ks_g = skinny_ks(2 * I, H, EPI_PARTIAL, B), ks_d = skinny_ks(H, I, EPI_PARTIAL, B);
    bf16_t *xa = c->dX, *xb = c->dX2;             // current / next residual rows
    for (int l = 0; l < c->NL; ++l) {
        const LlmLayer& W = c->llm[l];
        bf16_t* kc = c->kcache + (size_t)l * c->layer_stride;
        bf16_t* vc = c->vcache + (size_t)l * c->layer_stride;
        LCHK(launch_skinny_gemm_fused_norm(l ? c->sk_ws : nullptr, l ? ks_d : 0, xa, H, xb, H, W.rms1, c->c.rms_eps, W.wqkv_d, B, QKV, H, c->sk_ws2,
                                           c->sk_ws_floats, s));
        std::swap(xa, xb);
        LCHK(launch_attn_decode(c->dQKV, QKV, kc, vc, (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, c->d_slots, c->d_pos, c->dO,
                                H, c->attn_ws, c->tickets, B, c->NQ, c->NKV, HD, decode_nsplit(B), 1.0f / sqrtf((float)HD), 1,
                                c->rope_cos, c->rope_sin, c->sk_ws2, ks_q, s));
        LCHK(launch_skinny_gemm(c->dO, H, W.wo_d, H, nullptr, H, nullptr, 0, B, H, H, EPI_PARTIAL, 1, SKWS(c), s));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (l == 0 && c->profile == 2 && (c->bracket_mask & 2) && s != c->cap_stream && c->kev_used + 2 <= (int)c->kev.size()) {
            e0 = c->kev[c->kev_used]; e1 = c->kev[c->kev_used + 1]; c->kev_used += 2;
            c->bracket_kind = 2;
        }
        if (e0) hipEventRecord(e0, s);
        LCHK(launch_skinny_gemm_fused_norm(c->sk_ws, ks_o, xa, H, xb, H, W.rms2, c->c.rms_eps, W.wgu_d, B, 2 * I, H, c->sk_ws2, c->sk_ws_floats, s));
        if (e1) hipEventRecord(e1, s);
        std::swap(xa, xb);
        if (g_decode_fuse_swiglu) {        // SwiGLU folded into the down GEMV's parking step (round 4): 4 launches per layer
            LCHK(launch_skinny_gemm_fused_swiglu(c->sk_ws2, ks_g, W.wd_d, B, H, I, c->sk_ws, c->sk_ws_floats, s));
        } else {
            LCHK(launch_swiglu_combine(c->sk_ws2, ks_g, 2 * I, c->dACT, I, B, s));
            LCHK(launch_skinny_gemm(c->dACT, I, W.wd_d, I, nullptr, H, nullptr, 0, B, H, I, EPI_PARTIAL, 1, SKWS(c), s));
        }
    }
    // an even number of fused GEMVs: the residual rows are back in dX; the last layer's down partials + residual -> final norm -> heads
    LCHK(launch_add_rmsnorm(c->sk_ws, ks_d, xa, H, xa, H, c->final_norm, c->dH, H, B, H, c->c.rms_eps, s));
    return head_and_select(c, c->dH, 1, logits_out, s);
}

int g_decode_unfused = 0;   // RoPE + cache append as a kernel of its own before the decode attention: 0 = from batch 32 up (bit-identical to the fused
                            // prologue, 1 % faster per 64-sequence step, one launch more — which batch 1 would feel), 1 = always, 2 = never
                            // (trace_op_set_gemm_variant(120 + x), tools/decode_variant_ab.py --variants 122,121)
// one decode step for the current batch: consumes dX (embedding of the last token), leaves the next one in dX
static int decode_step(trace_ctx* c, float* logits_out, hipStream_t s) {
    const int H = c->H, I = c->I, HD = c->HD, QKV = c->QKV, B = c->B;
    // Every GEMV leaves fp32 k-chunk partial rows in sk_ws and its consumer sums them on load (an in-kernel merge costs
    // 5-8 us of dependent round trips per GEMV): qkv -> attention (RoPE + cache append + attention) -> o -> [sum + residual
    // -> new residual, RMSNorm] -> gate|up -> [sum, SwiGLU] -> down -> [sum + residual, next layer's / the final RMSNorm].
    if (B > SKINNY_ROWS || (B >= g_decode_wide_min && !c->fp8)) return decode_step_wide(c, logits_out, s);      // (fp8 contexts: at most 64 rows, checked at begin)
    if (B <= g_decode_fuse_norm_rows && !c->fp8 && skinny_fused_norm_ok(QKV, H, B) && skinny_fused_norm_ok(2 * I, H, B)) return decode_step_fused(c, logits_out, s);
    const bool wo = c->fp8 && c->fp8_wonly;          // weight-only decode GEMVs: bf16 activations straight from dH / dO / dACT, no quantiser launches
    const bool f8 = c->fp8 && !wo;
    auto ksf = [&](int N, int K) { return wo ? skinny_w8_ks(N, K, B) : f8 ? skinny_fp8_ks(N, K, B) : skinny_ks(N, K, EPI_PARTIAL, B); };
    const int ks_q = ksf(QKV, H), ks_o = ksf(H, H), ks_g = ksf(2 * I, H), ks_d = ksf(H, I);
#define GEMVW(X_, W8D_, SW_, N_, K_) LCHK(launch_skinny_w8((X_), (K_), (W8D_), (SW_), B, (N_), (K_), c->sk_ws, c->sk_ws_floats, s));
    // fp8: the GEMV's activations are quantised row-wise first (quant_rows_fp8: [B, K] bf16 -> e4m3 + per-row scale), the weights come
    // from the e4m3 tile copy; partial rows, their consumers and the attention are the bf16 path's
#define GEMV8(X_, W8D_, SW_, N_, K_)                                                                           \
    LCHK(launch_quant_rows_fp8((X_), (K_), c->dA8, (K_), c->dsa, B, (K_), s));                                  \
    LCHK(launch_skinny_fp8(c->dA8, (K_), c->dsa, (W8D_), (SW_), B, (N_), (K_), c->sk_ws, c->sk_ws_floats, s));
    // the normed hidden rows arrive already quantised (dH8 / dsh) from add_rmsnorm; only the step's first norm needs the quantiser
#define GEMV8H(W8D_, SW_, N_) LCHK(launch_skinny_fp8(c->dH8, H, c->dsh, (W8D_), (SW_), B, (N_), H, c->sk_ws, c->sk_ws_floats, s));
    LCHK(launch_rmsnorm(c->dX, H, c->dH, H, c->llm[0].rms1, B, H, c->c.rms_eps, s));
    if (f8) LCHK(launch_quant_rows_fp8(c->dH, H, c->dH8, H, c->dsh, B, H, s));
    for (int l = 0; l < c->NL; ++l) {
        const LlmLayer& W = c->llm[l];
        bf16_t* kc = c->kcache + (size_t)l * c->layer_stride;
        bf16_t* vc = c->vcache + (size_t)l * c->layer_stride;
        // (fusing the RMSNorm into the GEMV itself was tried: re-scaling the same activations in every workgroup cost
        //  more than a row kernel — 65 us vs 52 + 6 us for the gate|up GEMV)
        if (wo) { GEMVW(c->dH, W.wqkv8_d, W.sqkv, QKV, H) }
        else if (f8) { GEMV8H(W.wqkv8_d, W.sqkv, QKV) }
        else LCHK(launch_skinny_gemm(c->dH, H, W.wqkv_d, H, nullptr, QKV, nullptr, 0, B, QKV, H, EPI_PARTIAL, 1, SKWS(c), s));
        if (g_decode_unfused == 1 || (g_decode_unfused == 0 && B >= 32)) {
            LCHK(launch_qkv_finish(c->sk_ws, ks_q, QKV, c->dQKV, kc, vc, (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, c->d_slots,
                                   c->d_pos, B, c->NQ, c->NKV, c->rope_cos, c->rope_sin, s));
            LCHK(launch_attn_decode(c->dQKV, QKV, kc, vc, (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, c->d_slots, c->d_pos, c->dO,
                                    H, c->attn_ws, c->tickets, B, c->NQ, c->NKV, HD, decode_nsplit(B), 1.0f / sqrtf((float)HD), 0,
                                    nullptr, nullptr, nullptr, 0, s));
        } else
        LCHK(launch_attn_decode(c->dQKV, QKV, kc, vc, (long)c->slot_stride, (long)c->kv_head_stride, c->ctx_pad, c->d_slots, c->d_pos, c->dO,
                                H, c->attn_ws, c->tickets, B, c->NQ, c->NKV, HD, decode_nsplit(B), 1.0f / sqrtf((float)HD), 1,
                                c->rope_cos, c->rope_sin, c->sk_ws, ks_q, s));
        if (wo) { GEMVW(c->dO, W.wo8_d, W.so, H, H) }
        else if (f8) { GEMV8(c->dO, W.wo8_d, W.so, H, H) }
        else LCHK(launch_skinny_gemm(c->dO, H, W.wo_d, H, nullptr, H, nullptr, 0, B, H, H, EPI_PARTIAL, 1, SKWS(c), s));
        LCHK(launch_add_rmsnorm(c->sk_ws, ks_o, c->dX, H, c->dX, H, W.rms2, c->dH, H, B, H, c->c.rms_eps, s, f8 ? c->dH8 : nullptr, f8 ? c->dsh : nullptr));
        // roofline probe: HIP events around ONE launch of the dominant kernel (layer 0 gate|up GEMV) per step
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (l == 0 && c->profile == 2 && (c->bracket_mask & 2)) {
            // (event-record nodes captured into a hipGraph do not yield usable timestamps on ROCm 7.2: eager launches only)
            if (s != c->cap_stream && c->kev_used + 2 <= (int)c->kev.size()) { e0 = c->kev[c->kev_used]; e1 = c->kev[c->kev_used + 1]; c->kev_used += 2; c->bracket_kind = 1; }
        }
        if (e0) hipEventRecord(e0, s);
        if (wo) { GEMVW(c->dH, W.wgu8_d, W.sgu, 2 * I, H) }
        else if (f8) { GEMV8H(W.wgu8_d, W.sgu, 2 * I) }
        else LCHK(launch_skinny_gemm(c->dH, H, W.wgu_d, H, nullptr, 2 * I, nullptr, 0, B, 2 * I, H, EPI_PARTIAL, 1, SKWS(c), s));
        if (e1) hipEventRecord(e1, s);
        // (SwiGLU inside the gate|up GEMV at batch <= 16 — one K chunk, no combine kernel — measured 3.73 vs 3.5 ms per batch-1 step: its 224
        //  workgroups leave 32 CUs without a weight stream)
        LCHK(launch_swiglu_combine(c->sk_ws, ks_g, 2 * I, c->dACT, I, B, s));
        if (wo) { GEMVW(c->dACT, W.wd8_d, W.sd, H, I) }
        else if (f8) { GEMV8(c->dACT, W.wd8_d, W.sd, H, I) }
        else LCHK(launch_skinny_gemm(c->dACT, I, W.wd_d, I, nullptr, H, nullptr, 0, B, H, I, EPI_PARTIAL, 1, SKWS(c), s));
        const bf16_t* nw = l + 1 < c->NL ? c->llm[l + 1].rms1 : c->final_norm;
        LCHK(launch_add_rmsnorm(c->sk_ws, ks_d, c->dX, H, c->dX, H, nw, c->dH, H, B, H, c->c.rms_eps, s, f8 ? c->dH8 : nullptr, f8 ? c->dsh : nullptr));
    }
#undef GEMV8
#undef GEMV8H
#undef GEMVW
    return head_and_select(c, c->dH, 1, logits_out, s);
}

extern "C" int trace_decode_begin(trace_ctx* c, const int32_t* slots, int B, const int32_t* heads, int max_new, int eos,
                                  const int32_t* forced, float* logits_out, void* stream) {
    if (!c || !c->finalized) return fail(TRACE_ERR_STATE, "context not finalized");
    if (!slots || !heads || B < 1 || B > c->max_B || B > (c->fp8 ? SKINNY_ROWS : SK_ROWS)) return fail(TRACE_ERR_ARG, "bad batch (at most " + std::to_string(c->fp8 ? SKINNY_ROWS : SK_ROWS) + " sequences decode together)");
    if (max_new < 1 || max_new > c->c.max_new_tokens) return fail(TRACE_ERR_ARG, "max_new exceeds capacity");
    hipStream_t s = (hipStream_t)stream;
    int32_t pos[SK_ROWS], zero[SK_ROWS] = {0};
    for (int b = 0; b < B; ++b) {
        if (slots[b] < 0 || slots[b] >= c->max_B || c->slot_len[slots[b]] <= 0) return fail(TRACE_ERR_STATE, "slot not prefilled");
        if (heads[b] < 0 || heads[b] > 2) return fail(TRACE_ERR_ARG, "head must be 0, 1 or 2");
        pos[b] = c->slot_len[slots[b]];
        if (pos[b] + max_new > c->max_ctx) return fail(TRACE_ERR_ARG, "prefill + max_new_tokens exceeds max_ctx");
        for (int b2 = 0; b2 < b; ++b2) if (slots[b2] == slots[b]) return fail(TRACE_ERR_ARG, "duplicate slot");
    }
    c->B = B; c->max_new = max_new; c->eos = eos; c->has_forced = forced != nullptr;
    c->pos_sum = 0;
    for (int b = 0; b < B; ++b) c->pos_sum += pos[b];
    HIPCHK(hipMemcpyAsync(c->d_slots, slots, B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_pos, pos, B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_heads, heads, B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_done, zero, B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_out_len, zero, B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_step, zero, 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->d_out_ids, 0, (size_t)B * max_new * 4, s));
    const int32_t prm[3] = {max_new, eos, c->host_mode};
    HIPCHK(hipMemcpyAsync(c->d_params, prm, 12, hipMemcpyHostToDevice, s));
    c->fed = 0; c->steps_done = 0;
    if (forced) HIPCHK(hipMemcpyAsync(c->d_forced, forced, (size_t)B * max_new * 4, hipMemcpyHostToDevice, s));
    else HIPCHK(hipMemsetAsync(c->d_forced, 0xff, (size_t)B * max_new * 4, s));      // -1 = not forced
    // gather the prefill hidden rows of the chosen slots into dH, then head + select (no position advance)
    for (int b = 0; b < B; ++b)
        HIPCHK(hipMemcpyAsync(c->dH + (size_t)b * c->H, c->xlast + (size_t)slots[b] * c->H, (size_t)c->H * 2, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));     // host stack arrays above must outlive the async copies
    return head_and_select(c, c->dH, 0, logits_out, s);
}

extern "C" int trace_decode_steps(trace_ctx* c, int n, int use_graph, float* logits_out, void* stream) {
    if (!c || c->B < 1) return fail(TRACE_ERR_STATE, "trace_decode_begin not called");
    if (n < 0) return fail(TRACE_ERR_ARG, "bad n");
    if (logits_out && (n != 1 || use_graph)) return fail(TRACE_ERR_ARG, "logits_out needs n == 1 and eager mode");
    if (c->host_mode && (n != 1 || use_graph)) return fail(TRACE_ERR_ARG, "host-select mode runs one eager step at a time");
    // every step appends one KV row at pos[b]++; decode_begin checked pos + max_new <= max_ctx once, so the total is bounded here
    if (c->steps_done + n > c->max_new - 1)
        return fail(TRACE_ERR_STATE, "decode steps exceed max_new - 1 since trace_decode_begin (" + std::to_string(c->steps_done) + " taken, " +
                                     std::to_string(n) + " requested, max_new " + std::to_string(c->max_new) + ")");
    const int steps_before = c->steps_done;
    c->steps_done += n;
    hipStream_t s = (hipStream_t)stream;
    if (c->profile) hipEventRecord(c->ev0, s);
    if (!use_graph) {
        for (int i = 0; i < n; ++i) { c->step_in_call = steps_before + i; TRY(decode_step(c, logits_out, s)); }
    } else {
        const int key = c->B;
        hipGraphExec_t* slot_g = &c->graphs[key];
        if (!*slot_g) {
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));   // another thread may be launching the encode stage (generate_stream)
            const int rc = decode_step(c, nullptr, c->cap_stream);
            hipError_t e = hipStreamEndCapture(c->cap_stream, &g);
            if (rc != TRACE_OK) { if (g) hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) return fail(TRACE_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            e = hipGraphInstantiate(slot_g, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            if (e != hipSuccess) { *slot_g = nullptr; return fail(TRACE_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
        }
        for (int i = 0; i < n; ++i) HIPCHK(hipGraphLaunch(*slot_g, s));
    }
    if (c->profile == 2 && c->kev_used > 0) {   // drain the eager-mode brackets
        hipStreamSynchronize(s);
        for (int i = 0; i + 1 < c->kev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->kev[i], c->kev[i + 1]) == hipSuccess) { c->ksum_ms += ms; c->ksamples += 1; }
        }
        c->kev_used = 0;
    }
    if (c->profile) {
        hipEventRecord(c->ev1, s);
        hipEventSynchronize(c->ev1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->prof[0] = n > 0 ? ms / n : 0.f;
        c->prof[1] = (float)n;
        c->prof[2] = c->ksamples ? (float)(c->ksum_ms / c->ksamples) : 0.f;
        c->prof[3] = (float)c->ksamples;
        // algorithmic bytes of the bracketed launch, by WHICH launch took the bracket (bracket_kind — not by the batch size: the wide step starts at
        // g_decode_wide_min rows, below SKINNY_ROWS): the gate|up weights (fp8: the bracket also spans the activation quantiser), or the layer-0
        // attention's KV rows averaged over the bracketed steps
        c->prof[4] = (float)(2.0 * c->I * c->H * (c->fp8 ? 1.0 : 2.0));
        if (c->bracket_kind == 3 && c->ksamples) c->prof[4] = (float)(c->kbytes_sum / c->ksamples);
        c->prof[8] = (float)c->bracket_kind;
    }
    return TRACE_OK;
}

extern "C" int trace_decode_read(trace_ctx* c, int32_t* out_ids, int32_t* out_len, int32_t* heads, void* stream) {
    if (!c || c->B < 1) return fail(TRACE_ERR_STATE, "trace_decode_begin not called");
    hipStream_t s = (hipStream_t)stream;
    if (out_ids) HIPCHK(hipMemcpyAsync(out_ids, c->d_out_ids, (size_t)c->B * c->max_new * 4, hipMemcpyDeviceToHost, s));
    if (out_len) HIPCHK(hipMemcpyAsync(out_len, c->d_out_len, c->B * 4, hipMemcpyDeviceToHost, s));
    if (heads) HIPCHK(hipMemcpyAsync(heads, c->d_heads, c->B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return TRACE_OK;
}

extern "C" int trace_decode_host_mode(trace_ctx* c, int on) {
    if (!c) return fail(TRACE_ERR_ARG, "null ctx");
    c->host_mode = on ? 1 : 0;
    return TRACE_OK;
}

extern "C" int trace_decode_feed(trace_ctx* c, const int32_t* tokens, int B, void* stream) {
    if (!c || c->B < 1 || !c->host_mode) return fail(TRACE_ERR_STATE, "trace_decode_feed needs host-select mode and an active batch");
    if (!tokens || B != c->B) return fail(TRACE_ERR_ARG, "bad tokens / B");
    if (c->fed >= c->max_new) return fail(TRACE_ERR_STATE, "max_new tokens already fed");
    hipStream_t s = (hipStream_t)stream;
    for (int b = 0; b < B; ++b) {
        if (tokens[b] < 0 || tokens[b] >= c->NV) return fail(TRACE_ERR_ARG, "token id out of range");
        HIPCHK(hipMemcpyAsync(c->d_forced + (size_t)b * c->max_new + c->fed, tokens + b, 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    TRY(select_only(c, c->fed > 0 ? 1 : 0, s));
    c->fed += 1;
    return TRACE_OK;
}

// ------------------------------------------------------------------------------------------------ pipeline streams
extern int g_gemm_pers_grid_cap;
extern "C" int trace_stream_create(trace_ctx* c, int cu_first, int cu_count, void** stream_out) {
    if (!c || !stream_out) return fail(TRACE_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->dev));
    hipStream_t s = nullptr;
    if (cu_count == 0) {
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, c->dev));
        const int ncu = prop.multiProcessorCount;
        if (cu_first < 0 || cu_count < 8 || (cu_first % 8) || (cu_count % 8) || cu_first + cu_count > ncu)
            return fail(TRACE_ERR_ARG, "CU range must be multiples of 8 (one CU per XCD) inside the device's " + std::to_string(ncu) + " CUs");
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int i = cu_first; i < cu_first + cu_count; ++i) mask[i / 32] |= 1u << (i % 32);
        HIPCHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    }
    if (gemm_pers_init(s) != TRACE_OK) { hipStreamDestroy(s); return fail(TRACE_ERR_HIP, "persistent GEMM ticket counters"); }
    if (cu_count > 0) gemm_pers_set_cap(s, cu_count);          // a CU-masked stream: its persistent GEMMs launch one workgroup per CU IT can use
    c->streams.push_back(s);
    *stream_out = (void*)s;
    return TRACE_OK;
}
extern "C" int trace_stream_destroy(trace_ctx* c, void* stream) {
    if (!c || !stream) return fail(TRACE_ERR_ARG, "null argument");
    auto it = std::find(c->streams.begin(), c->streams.end(), (hipStream_t)stream);
    if (it == c->streams.end()) return fail(TRACE_ERR_ARG, "not a stream of this context");
    HIPCHK(hipSetDevice(c->dev));
    HIPCHK(hipStreamSynchronize(*it));
    gemm_pers_forget(*it);
    HIPCHK(hipStreamDestroy(*it));
    c->streams.erase(it);
    return TRACE_OK;
}
extern "C" int trace_set_gemm_cus(trace_ctx* c, int n) {
    if (!c || n < 0) return fail(TRACE_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->dev));
    // per stream, and only this context's streams (trace_stream_create already sets a CU-masked stream's cap to its CU count; this overrides it).
    // Until round 3 this wrote a process-wide number that nothing reset: every later persistent GEMM of the process stayed capped.
    for (hipStream_t st : c->streams)
        if (gemm_pers_set_cap(st, n) != TRACE_OK) return fail(TRACE_ERR_HIP, "persistent GEMM ticket counters");
    return TRACE_OK;
}

// debugging aid (tools/pipeline_stress.py): device addresses and element strides of the KV caches and of the prefill's last-position hidden rows
extern "C" int trace_debug_buffers(trace_ctx* c, void** kcache, void** vcache, void** xlast, int64_t* strides) {
    if (!c || !kcache || !vcache || !xlast || !strides) return fail(TRACE_ERR_ARG, "null argument");
    *kcache = c->kcache; *vcache = c->vcache; *xlast = c->xlast;
    strides[0] = (int64_t)c->layer_stride; strides[1] = (int64_t)c->slot_stride; strides[2] = (int64_t)c->kv_head_stride; strides[3] = c->ctx_pad;
    strides[4] = c->NL; strides[5] = c->NKV; strides[6] = c->HD; strides[7] = c->H;
    return TRACE_OK;
}

extern "C" int trace_set_profile(trace_ctx* c, int on) {
    if (!c) return fail(TRACE_ERR_ARG, "null ctx");
    c->profile = on; c->bracket_kind = 0; c->ksum_ms = 0.0; c->ksamples = 0; c->kev_used = 0; c->kbytes_sum = 0.0; c->msum_ms = 0.0; c->msamples = 0; c->mM = 0;
    return TRACE_OK;
}
extern "C" int trace_set_profile_brackets(trace_ctx* c, int mask) {
    if (!c) return fail(TRACE_ERR_ARG, "null ctx");
    c->bracket_mask = mask & 3;
    return TRACE_OK;
}
extern "C" int trace_get_profile(trace_ctx* c, float* out, int n) {
    if (!c || !out) return fail(TRACE_ERR_ARG, "null argument");
    for (int i = 0; i < n && i < 20; ++i) out[i] = c->prof[i];
    return TRACE_OK;
}

// ------------------------------------------------------------------------------------------------ op-level hooks
extern "C" int trace_op_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* R,
                             int ldr, int M, int N, int K, int epilogue, void* stream) {
    return gemm((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (bf16_t*)C, ldc, (const bf16_t*)bias, (const bf16_t*)R, ldr, M, N, K,
                epilogue, (hipStream_t)stream);
}
extern int g_gemm_variant;
extern int g_attn_debug;
extern int g_attn_pf_debug;
extern int g_skinny_debug;
extern int g_gemm_pers_opt;
extern int g_gemm_ldr_opt;
extern int g_gemm_pers_walk;
extern int g_gemm_resid_pers;
extern int g_gemm_w4;
extern int g_gemm_w4_opt;
extern int g_attn_vit_big;
extern int g_partial_cfg;
extern int g_attn_decode_w3;
extern int g_attn_decode_nt;
extern int g_attn_decode_lds_pad;
extern int g_partial_wgs;
extern "C" int trace_op_set_gemm_variant(int variant) {
    if (variant >= 1000 && variant <= 1000 + 1024) { g_gemm_pers_grid_cap = variant - 1000; return TRACE_OK; }   // persistent GEMM: at most n workgroups (0 = #CUs)
    if (variant >= 100 && variant < 110) { g_attn_debug = variant - 100; return TRACE_OK; }      // decode attention knock-outs (108: V^T read as contiguous blocks, timing only)
    if (variant >= 110 && variant < 120) { g_attn_pf_debug = variant - 110; return TRACE_OK; }
    if (variant >= 120 && variant <= 122) { g_decode_unfused = variant - 120; return TRACE_OK; }
    if (variant >= 130 && variant <= 137) { g_decode_gemm_tiled = variant - 130; return TRACE_OK; }
    if (variant >= 760 && variant <= 764) { g_attn_decode_w3 = variant == 762 ? -1 : variant - 760; return TRACE_OK; }
    if (variant >= 780 && variant <= 799) { g_attn_decode_lds_pad = (variant - 780) * 8; return TRACE_OK; }
    if (variant >= 770 && variant <= 771) { g_attn_decode_nt = variant - 770; return TRACE_OK; }
    if (variant >= 750 && variant <= 751) { g_prefill_last_rows = variant - 750; return TRACE_OK; }
    if (variant >= 740 && variant <= 743) { g_partial_cfg = variant - 740; return TRACE_OK; }        // tile shape of the decode partial-row GEMM (gemm.hip)
    if (variant >= 800 && variant <= 832) { g_partial_wgs = (variant - 800) * 32; return TRACE_OK; }   // its workgroup target (0 = default 256)
    if (variant >= 700 && variant < 732) { g_decode_gemm_tiled = variant - 700; return TRACE_OK; }   // the same word with its round-6 bits: 2 = nt weight DMA, 8 / 16 = nt / write-through partial-row stores
    if (variant >= 160 && variant <= 161) { g_vit_patch_fused = variant - 160; return TRACE_OK; }
    if (variant >= 170 && variant <= 174) { g_decode_fuse_norm_rows = variant - 170; return TRACE_OK; }
    if (variant >= 180 && variant <= 181) { g_decode_fuse_swiglu = variant - 180; return TRACE_OK; }
    if (variant >= 190 && variant <= 192) { g_attn_vit_big = variant - 190; return TRACE_OK; }   // ViT attention: 0 = the 4 x 32-row kernel, 1 = the 192-row kernel (4-stage ring), 2 = (3-stage ring)
    if (variant >= 850 && variant <= 866) { g_attn_decode_nsplit = variant - 850; return TRACE_OK; }
    if (variant >= 144 && variant <= 145) { g_decode_wide_fuse_qkv = variant - 144; return TRACE_OK; }
    if (variant >= 140 && variant <= 143) { g_decode_wide_min = variant == 140 ? SKINNY_ROWS + 1 : variant == 141 ? 33 : variant == 142 ? 17 : 32; return TRACE_OK; }
    if (variant >= 200 && variant < 210) { g_skinny_debug = variant - 200; return TRACE_OK; }
    if (variant >= 300 && variant < 364) { g_gemm_pers_opt = variant - 300; return TRACE_OK; }
    if (variant >= 500 && variant <= 501) { g_gemm_pers_walk = variant - 500; return TRACE_OK; }   // the persistent GEMM's tile walk on every route (gemm_pers.hip)
    if (variant >= 520 && variant <= 521) { g_gemm_resid_pers = variant - 520; return TRACE_OK; }
    if (variant >= 530 && variant <= 531) { g_gemm_w4 = variant - 530; return TRACE_OK; }
    if (variant >= 540 && variant < 548) { g_gemm_w4_opt = variant - 540; return TRACE_OK; }   // gemm_w4.hip A/B builds: bit 0 = one barrier per K-tile, bit 2 = L2 touches of the A panel   // non-residual 256^2 shapes: 0 = gemm_pers.hip, 1 = gemm_w4.hip (auto routing)   // residual GEMMs on the persistent kernel too (auto routing)
    if (variant >= 400 && variant < 404) { g_gemm_ldr_opt = variant - 400; return TRACE_OK; }   // gemm_ldr A/B: bit 0 = no residual touches, bit 1 = no A-panel touches
    if (variant < 0 || variant > 8) return fail(TRACE_ERR_ARG, "variant must be 0..8");
    g_gemm_variant = variant;
    return TRACE_OK;
}
extern "C" int trace_op_layernorm(const void* x, void* y, const void* w, const void* b, int rows, int D, float eps, void* stream) {
    LCHK(launch_layernorm((const bf16_t*)x, D, (bf16_t*)y, D, (const bf16_t*)w, (const bf16_t*)b, rows, D, eps, (hipStream_t)stream));
    return TRACE_OK;
}
extern "C" int trace_op_rmsnorm(const void* x, void* y, const void* w, int rows, int D, float eps, void* stream) {
    LCHK(launch_rmsnorm((const bf16_t*)x, D, (bf16_t*)y, D, (const bf16_t*)w, rows, D, eps, (hipStream_t)stream));
    return TRACE_OK;
}
// Q [batch, nq, heads, hd], K/V [batch, nkv, kv_heads, hd] (token-major, heads interleaved), O like Q.
// vt_scratch: batch*kv_heads*hd*round_up(nkv,64) bf16.
extern "C" int trace_op_attention(const void* Q, const void* K, const void* V, void* O, void* vt_scratch, int batch, int heads,
                                  int kv_heads, int nq, int nkv, int head_dim, int causal, float scale, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int pad = round_up(nkv, 64);
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)vt_scratch; a.O = (bf16_t*)O;
    a.q_bs = (long)nq * heads * head_dim; a.q_hs = head_dim; a.q_rs = heads * head_dim;
    a.k_bs = (long)nkv * kv_heads * head_dim; a.k_hs = head_dim; a.k_rs = kv_heads * head_dim;
    a.v_bs = (long)kv_heads * head_dim * pad; a.v_hs = (long)head_dim * pad; a.v_rs = pad;
    a.o_bs = a.q_bs; a.o_hs = head_dim; a.o_rs = heads * head_dim;
    a.nq_rows = nq; a.nkv_rows = nkv; a.batch = batch; a.heads = heads; a.kv_heads = kv_heads; a.scale = scale; a.causal = causal;
    a.Vrow = (const bf16_t*)V; a.vr_bs = a.k_bs; a.vr_hs = head_dim; a.vr_rs = kv_heads * head_dim;
    a.v_perm = (head_dim == 64 && !causal && heads == kv_heads && attn_vit_wants_perm(nkv, true)) ? (attn_vit_rowmajor_v() ? 2 : 1) : 0;
    if (a.v_perm != 2)
        LCHK(launch_transpose_v((const bf16_t*)V, a.k_bs, head_dim, kv_heads * head_dim, (bf16_t*)vt_scratch, a.v_bs, a.v_hs, pad, nkv,
                                head_dim, kv_heads, batch, s, a.v_perm));
    if (head_dim == 64) { LCHK(launch_attn_vit(a, s)); }
    else if (head_dim == 128) { LCHK(launch_attn_prefill(a, s)); }
    else return fail(TRACE_ERR_ARG, "head_dim must be 64 or 128");
    return TRACE_OK;
}
// Decode GEMV test hooks.  w_tiled: W is in the decode tile layout (trace_op_tile_pack) instead of row-major [N][K].
// epilogue 4 (EPI_PARTIAL): `out` receives the fp32 partial rows [trace_op_skinny_ks(...)][64][N] instead of bf16.
static float* g_sk_ws = nullptr;
static unsigned int* g_sk_tk = nullptr;
static size_t g_sk_ws_floats = 0;
static int g_sk_ntk = 0;
extern "C" int trace_op_skinny_gemm(const void* X, const void* W, void* out, const void* R, int B, int N, int K, int epilogue,
                                    int w_tiled, void* stream) {
    const int No = epilogue == EPI_SWIGLU ? N / 2 : N;
    const size_t need = skinny_ws_floats(N, K, epilogue);
    if (need > g_sk_ws_floats || N / 16 > g_sk_ntk) {             // grow-only scratch (never freed)
        HIPCHK(hipDeviceSynchronize());
        if (g_sk_ws) hipFree(g_sk_ws);
        if (g_sk_tk) hipFree(g_sk_tk);
        g_sk_ws_floats = std::max(need, g_sk_ws_floats) + 64; g_sk_ntk = std::max(N / 16, g_sk_ntk);
        HIPCHK(hipMalloc((void**)&g_sk_ws, g_sk_ws_floats * 4));
        HIPCHK(hipMalloc((void**)&g_sk_tk, (size_t)g_sk_ntk * 4));
        HIPCHK(hipMemset(g_sk_tk, 0, (size_t)g_sk_ntk * 4));
    }
    LCHK(launch_skinny_gemm((const bf16_t*)X, K, (const bf16_t*)W, K, (bf16_t*)out, No, (const bf16_t*)R, No, B, N, K, epilogue,
                            w_tiled, g_sk_ws, g_sk_ws_floats, g_sk_tk, g_sk_ntk, (hipStream_t)stream));
    if (epilogue == EPI_PARTIAL && out)
        HIPCHK(hipMemcpyAsync(out, g_sk_ws, (size_t)skinny_ks(N, K, epilogue, B) * SK_ROWS * N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return TRACE_OK;
}
extern "C" int trace_op_skinny_ks(int N, int K, int epilogue, int B) { return skinny_ks(N, K, epilogue, B); }
extern "C" int trace_op_sk_rows(void) { return SK_ROWS; }
// gate|up of a wide decode step: X [M <= 128, K] . Wt (the 16-row interleaved gate|up matrix in the decode tile layout) -> SwiGLU -> out [M, N/2] bf16
extern "C" int trace_op_gemm_swiglu_tiled(const void* X, const void* Wt, void* out, int M, int N, int K, int ring, void* stream) {
    GemmArgs g{(const bf16_t*)X, K, (const bf16_t*)Wt, K, (bf16_t*)out, N / 2, nullptr, nullptr, 0, M, N, K, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, ring ? 5 : 1};
    const int rc = launch_gemm_bf16(g, EPI_SWIGLU, (hipStream_t)stream);
    if (rc != TRACE_OK) return fail(rc, "tiled SwiGLU GEMM launch failed");
    return TRACE_OK;
}
// [M <= 128, K] x [N, K]^T as fp32 k-chunk partial rows [trace_op_gemm_partial_ks(N, K)][trace_op_sk_rows()][N] (decode batches above 64 rows)
extern "C" int trace_op_gemm_partial_ks(int N, int K) { return gemm_partial_ks(N, K); }
extern "C" int trace_op_gemm_partial(const void* A, const void* W, float* part, int M, int N, int K, int w_tiled, void* stream) {
    GemmArgs g{(const bf16_t*)A, K, (const bf16_t*)W, K, nullptr, 0, nullptr, nullptr, 0, M, N, K, nullptr, 0, nullptr, nullptr, 0, part, gemm_partial_ks(N, K), w_tiled};
    const int rc = launch_gemm_bf16(g, EPI_PARTIAL, (hipStream_t)stream);
    if (rc != TRACE_OK) return fail(rc, "partial GEMM launch failed");
    return TRACE_OK;
}
// ---- fp8 path hooks (tests/test_gpu_fp8.py) ----
extern "C" int trace_op_quant_rows_fp8(const void* X, void* X8, float* sx, int rows, int K, void* stream) {
    LCHK(launch_quant_rows_fp8((const bf16_t*)X, K, (uint8_t*)X8, K, sx, rows, K, (hipStream_t)stream));
    return TRACE_OK;
}
extern "C" int trace_op_gemm_fp8(const void* A8, const float* sa, const void* W8, const float* sw, void* C, const void* R, int M, int N, int K,
                                 int epilogue, void* stream) {
    const int No = epilogue == EPI_SWIGLU ? N / 2 : N;
    GemmArgs g{(const bf16_t*)A8, K, (const bf16_t*)W8, K, (bf16_t*)C, No, nullptr, (const bf16_t*)R, No, M, N, K, nullptr, 1, sa, sw};
    const int rc = launch_gemm_bf16(g, epilogue, (hipStream_t)stream);
    if (rc != TRACE_OK) return fail(rc, "fp8 gemm launch failed");
    return TRACE_OK;
}
// X8 [B,K] e4m3 + sx, W8 [N,K] e4m3 row-major + sw -> out fp32 [B,N] (the k-chunk partial rows summed here, in chunk order)
extern "C" int trace_op_skinny_fp8(const void* X8, const float* sx, const void* W8, const float* sw, float* out, int B, int N, int K, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    static uint8_t* wt = nullptr; static size_t wt_bytes = 0;
    static float* ws = nullptr; static size_t ws_floats = 0;
    const size_t need_w = (size_t)N * K, need_ws = (size_t)skinny_fp8_ks(N, K, B) * SK_ROWS * N;
    if (need_w > wt_bytes) { HIPCHK(hipDeviceSynchronize()); if (wt) hipFree(wt); HIPCHK(hipMalloc((void**)&wt, need_w)); wt_bytes = need_w; }
    if (need_ws > ws_floats) { HIPCHK(hipDeviceSynchronize()); if (ws) hipFree(ws); HIPCHK(hipMalloc((void**)&ws, need_ws * 4)); ws_floats = need_ws; }
    LCHK(launch_tile_pack_fp8((const uint8_t*)W8, K, wt, N, K, s));
    LCHK(launch_skinny_fp8((const uint8_t*)X8, K, sx, wt, sw, B, N, K, ws, ws_floats, s));
    const int KS = skinny_fp8_ks(N, K, B);
    std::vector<float> h((size_t)KS * SK_ROWS * N), o((size_t)B * N, 0.f);
    HIPCHK(hipMemcpyAsync(h.data(), ws, h.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int ks = 0; ks < KS; ++ks)
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < N; ++n) o[(size_t)b * N + n] += h[((size_t)ks * SK_ROWS + b) * N + n];
    HIPCHK(hipMemcpy(out, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    return TRACE_OK;
}

// weight-only form: X bf16 [B,K], W8 [N,K] e4m3 row-major + sw -> out fp32 [B,N] (the k-chunk partial rows summed here, in chunk order)
extern "C" int trace_op_skinny_w8(const void* X, const void* W8, const float* sw, float* out, int B, int N, int K, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    static uint8_t* wt = nullptr; static size_t wt_bytes = 0;
    static float* ws = nullptr; static size_t ws_floats = 0;
    const int KS = skinny_w8_ks(N, K, B);
    const size_t need_w = (size_t)N * K, need_ws = (size_t)KS * SK_ROWS * N;
    if (need_w > wt_bytes) { HIPCHK(hipDeviceSynchronize()); if (wt) hipFree(wt); HIPCHK(hipMalloc((void**)&wt, need_w)); wt_bytes = need_w; }
    if (need_ws > ws_floats) { HIPCHK(hipDeviceSynchronize()); if (ws) hipFree(ws); HIPCHK(hipMalloc((void**)&ws, need_ws * 4)); ws_floats = need_ws; }
    LCHK(launch_tile_pack_fp8((const uint8_t*)W8, K, wt, N, K, s));
    LCHK(launch_skinny_w8((const bf16_t*)X, K, wt, sw, B, N, K, ws, ws_floats, s));
    std::vector<float> h((size_t)KS * SK_ROWS * N), o((size_t)B * N, 0.f);
    HIPCHK(hipMemcpyAsync(h.data(), ws, h.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int ks = 0; ks < KS; ++ks)
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < N; ++n) o[(size_t)b * N + n] += h[((size_t)ks * SK_ROWS + b) * N + n];
    HIPCHK(hipMemcpy(out, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    return TRACE_OK;
}

// the fused-norm decode GEMV of small batches: part_in [ks_in][sk_rows][K] fp32 + R [B,K] -> xout [B,K] (new residual), out = fp32 partial rows
// [trace_op_skinny_ks(N,K,4,B)][sk_rows][N] of RMSNorm(xout; w) . W^T (W row-major here; packed internally)
extern "C" int trace_op_skinny_fused_norm(const float* part_in, int ks_in, const void* R, void* xout, const void* w, float eps, const void* W, float* out,
                                          int B, int N, int K, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!skinny_fused_norm_ok(N, K, B)) return fail(TRACE_ERR_ARG, "shape not supported by the fused-norm GEMV");
    bf16_t* wt = nullptr; float* ws = nullptr;
    const size_t wsf = skinny_ws_floats(N, K, EPI_PARTIAL);
    HIPCHK(hipMalloc((void**)&wt, (size_t)N * K * 2)); HIPCHK(hipMalloc((void**)&ws, wsf * 4));
    int rc = launch_tile_pack((const bf16_t*)W, K, wt, N, K, s);
    if (rc == TRACE_OK) rc = launch_skinny_gemm_fused_norm(part_in, ks_in, (const bf16_t*)R, K, (bf16_t*)xout, K, (const bf16_t*)w, eps, wt, B, N, K, ws, wsf, s);
    if (rc == TRACE_OK && hipMemcpyAsync(out, ws, (size_t)skinny_ks(N, K, EPI_PARTIAL, B) * SK_ROWS * N * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = TRACE_ERR_HIP;
    hipStreamSynchronize(s);
    hipFree(wt); hipFree(ws);
    if (rc != TRACE_OK) return fail(rc, "fused-norm GEMV failed");
    return TRACE_OK;
}

extern "C" int trace_op_tile_pack(const void* W, void* Wt, int N, int K, void* stream) {
    LCHK(launch_tile_pack((const bf16_t*)W, K, (bf16_t*)Wt, N, K, (hipStream_t)stream));
    return TRACE_OK;
}
extern "C" int trace_op_swiglu_combine(const float* part, int KS, int N2, void* out, int B, void* stream) {
    LCHK(launch_swiglu_combine(part, KS, N2, (bf16_t*)out, N2 / 2, B, (hipStream_t)stream));
    return TRACE_OK;
}
extern "C" int trace_op_add_rmsnorm(const float* part, int KS, const void* R, void* xout, const void* w, void* y, int B, int N,
                                    float eps, void* stream) {
    LCHK(launch_add_rmsnorm(part, KS, (const bf16_t*)R, N, (bf16_t*)xout, N, (const bf16_t*)w, (bf16_t*)y, N, B, N, eps,
                            (hipStream_t)stream));
    return TRACE_OK;
}
// kcache [B, nkv, max_ctx, 128]; vtcache [B, nkv, 128, max_ctx] (V transposed; max_ctx % 32 == 0); pos[b] = index of the
// newest token (ctx = pos+1), already in the caches; q [B, nq*128] ready (rotated).  ws: B*nq*nsplit*130 floats.
extern "C" int trace_op_attn_decode(const void* q, const void* kcache, const void* vcache, const int32_t* pos, void* O, float* ws,
                                    int B, int nq, int nkv, int max_ctx, int nsplit, float scale, void* stream) {
    static int32_t* d_slots = nullptr;
    static unsigned int* d_tickets = nullptr;
    if (!d_slots) {
        int32_t h[SK_ROWS];
        for (int i = 0; i < SK_ROWS; ++i) h[i] = i;
        HIPCHK(hipMalloc((void**)&d_slots, SK_ROWS * 4));
        HIPCHK(hipMemcpy(d_slots, h, SK_ROWS * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&d_tickets, SK_ROWS * 64 * 4));
        HIPCHK(hipMemset(d_tickets, 0, SK_ROWS * 64 * 4));
    }
    LCHK(launch_attn_decode((const bf16_t*)q, nq * 128, (bf16_t*)kcache, (bf16_t*)vcache, (long)nkv * max_ctx * 128,
                            (long)max_ctx * 128, max_ctx, d_slots, pos, (bf16_t*)O, nq * 128, ws, d_tickets, B, nq, nkv, 128, nsplit, scale, 0,
                            nullptr, nullptr, nullptr, 0, (hipStream_t)stream));
    return TRACE_OK;
}
