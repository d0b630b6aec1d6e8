"""A/B of decode-step variants on the 7B engine (random embeddings prefilled, no ViT): the attention's fused prologue (RoPE + cache append inside
attn_decode_kernel: 122) against the same work as a kernel of its own (121; the engine's default from batch 32 up).  Interleaved rounds, median.
python tools/decode_ab.py [--batch B] [--ctx L]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=16)
a = ap.parse_args()
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=a.batch, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for b in range(a.batch):
    eng.prefill(b, a.ctx, embeds=emb)
torch.cuda.synchronize()
slots = list(range(a.batch))
lg = {}
for v in (0, 1):
    ops.set_gemm_variant(122 - v)                      # 122 = fused prologue, 121 = separate kernel
    steps = [eng.decode_begin(slots, [1] * a.batch, 256, eos=-1, want_logits=True).clone()]
    for _ in range(3):
        steps.append(eng.decode_steps(1, use_graph=False, want_logits=True).clone())
    lg[v] = torch.stack(steps)
print("logits of 4 steps identical:", torch.equal(lg[0], lg[1]), " max|d|", (lg[0] - lg[1])[torch.isfinite(lg[0])].abs().max().item())
ts = {0: [], 1: []}
for rnd in range(6):
    for v in (0, 1):
        ops.set_gemm_variant(122 - v)
        eng.decode_begin(slots, [1] * a.batch, 256, eos=-1)        # every round restarts at the prefilled context
        eng.decode_steps(2, use_graph=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_steps(a.steps, use_graph=False)
        torch.cuda.synchronize()
        ts[v].append((time.perf_counter() - t0) / a.steps * 1e3)
ops.set_gemm_variant(120)
for v, name in ((0, "fused prologue"), (1, "separate qkv_finish kernel")):
    print(f"batch {a.batch} ctx {a.ctx}: {name:28s} {statistics.median(ts[v][1:]):.3f} ms/step  (rounds: {' '.join('%.3f' % t for t in ts[v])})")
