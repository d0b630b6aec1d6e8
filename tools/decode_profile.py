"""Decode-only workload for rocprofv3: 7B engine, random embeddings prefilled (no ViT), then N decode steps.
python tools/decode_profile.py [--batch B] [--steps N] [--eager] [--ctx L]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--prefill-only", action="store_true")
a = ap.parse_args()
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=a.batch, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for b in range(a.batch):
    eng.prefill(b, a.ctx, embeds=emb)
torch.cuda.synchronize()
if not a.prefill_only:
    eng.decode_begin(list(range(a.batch)), [1] * a.batch, 256)
    eng.decode_steps(4, use_graph=not a.eager)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.decode_steps(a.steps, use_graph=not a.eager)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {a.batch} ctx {a.ctx}: {dt / a.steps * 1e3:.3f} ms/step  {a.batch * a.steps / dt:.1f} tok/s  ({'eager' if a.eager else 'graph'})")
