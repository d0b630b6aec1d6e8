"""A/B of the small-batch decode step on the 7B engine (random embeddings prefilled): add+RMSNorm as kernels of their own (variant 170) against the
fused-norm GEMVs (174: batches up to 4).  Interleaved rounds, median; logits compared.   python tools/decode_small_ab.py [--batch 1]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=48)
a = ap.parse_args()
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=a.batch, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for b in range(a.batch):
    eng.prefill(b, a.ctx, embeds=emb)
torch.cuda.synchronize()
slots = list(range(a.batch))
names = {0: "add+RMSNorm kernels", 4: "fused-norm GEMVs"}
lg = {}
for v in names:
    ops.set_gemm_variant(170 + v)
    steps = [eng.decode_begin(slots, [1] * a.batch, 256, eos=-1, want_logits=True).clone()]
    for _ in range(5):
        steps.append(eng.decode_steps(1, use_graph=False, want_logits=True).clone())
    lg[v] = torch.stack(steps)
fin = torch.isfinite(lg[0])
print(f"logits, fused vs unfused over 6 steps: max|d| {(lg[4][fin] - lg[0][fin]).abs().max().item():.4f}; finite pattern equal: {torch.equal(fin, torch.isfinite(lg[4]))}")
ts = {v: [] for v in names}
for rnd in range(5):
    for v in names:
        ops.set_gemm_variant(170 + v)
        for graph in (True,):
            eng.decode_begin(slots, [1] * a.batch, 256, eos=-1)
            eng.decode_steps(2, use_graph=False)          # (eager: a graph captured under one variant would replay it under the other)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.decode_steps(a.steps, use_graph=False)
            torch.cuda.synchronize()
            ts[v].append((time.perf_counter() - t0) / a.steps * 1e3)
ops.set_gemm_variant(174)
for v, name in names.items():
    m = statistics.median(ts[v][1:])
    print(f"batch {a.batch} ctx {a.ctx}: {name:24s} {m:.3f} ms/step = {a.batch / m * 1e3:.0f} tok/s  (rounds: {' '.join('%.3f' % t for t in ts[v])})")
