"""Batch-1 decode step (the reference drivers' call shape), A/B of the SwiGLU fold: swiglu_combine + down GEMV (variant 180) against the down GEMV with the
SwiGLU prologue (181, the default).  Two engines, each capturing its hipGraph under its own variant (a captured graph replays what it captured whatever
the switch says later); interleaved rounds of graph replays, median; logits compared bit for bit.   python tools/decode_b1_ab.py [--ctx 1968]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=64)
a = ap.parse_args()
cfg = tcfg.trace_7b()
names = {0: "swiglu_combine + down GEMV", 1: "down GEMV with SwiGLU prologue"}
engs, lg = {}, {}
torch.manual_seed(0)
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for v in names:
    ops.set_gemm_variant(180 + v)
    e = TraceEngine(cfg, max_batch=1, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
    e.load_weights(synth.iter_weights(cfg, device="cuda"))
    e.prefill(0, a.ctx, embeds=emb)
    steps = [e.decode_begin([0], [1], 256, eos=-1, want_logits=True).clone()]
    for _ in range(6):
        steps.append(e.decode_steps(1, use_graph=False, want_logits=True).clone())
    lg[v] = torch.stack(steps)
    e.decode_begin([0], [1], 256, eos=-1)
    e.decode_steps(4, use_graph=True)                  # captures this engine's batch-1 graph under variant v
    torch.cuda.synchronize()
    engs[v] = e
print("logits over 7 steps bit-identical:", torch.equal(lg[0], lg[1]))
ts = {v: [] for v in names}
for rnd in range(7):
    for v, e in engs.items():
        e.decode_begin([0], [1], 256, eos=-1)
        e.decode_steps(2, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.decode_steps(a.steps, use_graph=True)
        torch.cuda.synchronize()
        ts[v].append((time.perf_counter() - t0) / a.steps * 1e3)
ops.set_gemm_variant(181)
for v, name in names.items():
    m = statistics.median(ts[v][1:])
    print(f"batch 1 ctx {a.ctx}: {name:34s} {m:.3f} ms/step = {1e3 / m:.0f} tok/s  (rounds: {' '.join('%.3f' % t for t in ts[v])})")
