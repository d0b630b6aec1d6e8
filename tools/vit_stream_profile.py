"""The bench's ViT call shape for rocprofv3: three 170-frame tower calls (TraceEngine.full_round_frames) at the TRACE-7B
geometry, nothing else — so the kernel-trace average of gemm_pers_kernel<2, 0> (fc1, EPI_QUICKGELU) in this profile is the average
of exactly the launch shape bench.py brackets with HIP events (`roofline.avg_launch_ms`)."""
import dataclasses, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops
for v in os.environ.get("TRACE_VARIANTS", "").split(","):      # A/B switches (trace_op_set_gemm_variant), e.g. 150 = no LayerNorm fold, 160 = fold epilogue without prefetch
    if v.strip():
        ops.set_gemm_variant(int(v))
if os.environ.get("TRACE_ATTN_TRANSPOSED_V"):
    ops.set_gemm_variant(116)          # A/B: the round-2 path (transpose_v + permuted V^T) instead of row-major V through LDS transpose reads
cfg = dataclasses.replace(tcfg.trace_7b(128), num_hidden_layers=1)
eng = TraceEngine(cfg, max_batch=2, max_ctx=2304, max_frames=128, max_new_tokens=8)
eng.load_weights(synth.iter_weights(cfg, device="cuda:0"))
F = eng.vit_batch_frames
vids = [synth.synth_frames(cfg, b, num_frames=F, dtype=torch.bfloat16, device="cuda:0") for b in range(3)]
eng.vit_forward_many(vids)
torch.cuda.synchronize()
print("tower calls of", F, "frames")
eng.close()
