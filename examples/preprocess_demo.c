/* Plain-C client of the C ABI (include/trace_hip.h): what a non-Python host (the reference has none; a Go / Java / Rust
 * binding would look the same through cgo / JNI / FFI) needs to do to run one entry point of the path — here the frame
 * preprocessing of process_video (trace/mm_utils.py:456-462), which needs no weights.
 *
 *   gcc -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/preprocess_demo.c \
 *       -L trace_amd -ltrace_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/trace_amd -o preprocess_demo
 *   ./preprocess_demo 4 90 160          # T H W -> prints the output shape and a checksum of the bf16 result */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "trace_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_TRACE(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s: %s\n", #x, trace_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 4, H = argc > 2 ? atoi(argv[2]) : 90, W = argc > 3 ? atoi(argv[3]) : 160;
    printf("libtrace_hip ABI version %d\n", trace_abi_version());

    /* geometry only matters for v_image here; the LLM / ViT sizes just have to be self-consistent */
    trace_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.vocab_size = 320; cfg.hidden_size = 4096; cfg.intermediate_size = 256; cfg.num_layers = 1; cfg.num_heads = 32; cfg.num_kv_heads = 8;
    cfg.time_vocab = 13; cfg.score_vocab = 13; cfg.rms_eps = 1e-5f; cfg.rope_theta = 1e6f;
    cfg.v_hidden = 128; cfg.v_inter = 256; cfg.v_layers_used = 1; cfg.v_heads = 2; cfg.v_image = 56; cfg.v_patch = 14; cfg.v_eps = 1e-5f;
    cfg.num_slots = 8; cfg.slot_eps = 1e-6f; cfg.slot_rope_base = 10000.f;
    cfg.max_frames = T; cfg.max_ctx = 256; cfg.max_batch = 1; cfg.max_new_tokens = 8; cfg.projector_type = 0;
    trace_ctx* ctx = NULL;
    CHECK_TRACE(trace_ctx_create(&cfg, 0, &ctx));

    const size_t n_in = (size_t)T * H * W * 3, n_out = (size_t)T * 3 * cfg.v_image * cfg.v_image;
    uint8_t* h_in = (uint8_t*)malloc(n_in);
    for (size_t i = 0; i < n_in; ++i) h_in[i] = (uint8_t)((i * 2654435761u) >> 24);          /* deterministic pseudo-random pixels */
    void *d_in = NULL, *d_out = NULL;
    CHECK_HIP(hipMalloc(&d_in, n_in));
    CHECK_HIP(hipMalloc(&d_out, n_out * 2));
    CHECK_HIP(hipMemcpy(d_in, h_in, n_in, hipMemcpyHostToDevice));
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, std[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    CHECK_TRACE(trace_preprocess_frames(ctx, d_in, T, H, W, /*pad_to_square=*/1, mean, std, d_out, /*bf16*/0, /*stream*/NULL));
    CHECK_HIP(hipDeviceSynchronize());
    uint16_t* h_out = (uint16_t*)malloc(n_out * 2);
    CHECK_HIP(hipMemcpy(h_out, d_out, n_out * 2, hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (size_t i = 0; i < n_out; ++i) sum = sum * 1000003u + h_out[i];
    printf("out [%d,3,%d,%d] bf16 checksum %llu\n", T, cfg.v_image, cfg.v_image, (unsigned long long)sum);

    /* error convention: negative code + message, nothing thrown */
    if (trace_preprocess_frames(ctx, NULL, T, H, W, 1, mean, std, d_out, 0, NULL) >= 0) { fprintf(stderr, "expected an error\n"); return 4; }
    printf("error path: \"%s\"\n", trace_last_error());
    hipFree(d_in); hipFree(d_out); free(h_in); free(h_out);
    CHECK_TRACE(trace_ctx_destroy(ctx));
    return 0;
}
