"""A/B of the wide decode step's GEMM weight source on the 7B engine (random embeddings prefilled, no ViT): row-major prefill copies (130) vs decode
tile copies (131) vs tile copies + a 4-stage K-tile ring (135).  Interleaved rounds, median.   python tools/decode_gemm_ab.py [--batch 128]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--wide-min", type=int, default=0, help="1 / 2: batches from 33 / 17 up take the wide decode step (default: from 65)")
a = ap.parse_args()
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=a.batch, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for b in range(a.batch):
    eng.prefill(b, a.ctx, embeds=emb)
torch.cuda.synchronize()
slots = list(range(a.batch))
ops.set_gemm_variant(140 + a.wide_min)
names = {0: "row-major weights", 1: "decode tile copies", 5: "tile copies, 4-stage ring"}
lg = {}
for v in names:
    ops.set_gemm_variant(130 + v)
    steps = [eng.decode_begin(slots, [1] * a.batch, 256, eos=-1, want_logits=True).clone()]
    for _ in range(3):
        steps.append(eng.decode_steps(1, use_graph=False, want_logits=True).clone())
    lg[v] = torch.stack(steps)
fin = torch.isfinite(lg[0])
for v in (1, 5):
    print(f"logits vs row-major, {names[v]}: max|d| {(lg[v][fin] - lg[0][fin]).abs().max().item():.4f}")
ts = {v: [] for v in names}
for rnd in range(5):
    for v in names:
        ops.set_gemm_variant(130 + v)
        eng.decode_begin(slots, [1] * a.batch, 256, eos=-1)
        eng.decode_steps(2, use_graph=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_steps(a.steps, use_graph=False)
        torch.cuda.synchronize()
        ts[v].append((time.perf_counter() - t0) / a.steps * 1e3)
ops.set_gemm_variant(135)
for v, name in names.items():
    print(f"batch {a.batch} ctx {a.ctx}: {name:28s} {statistics.median(ts[v][1:]):.3f} ms/step  (rounds: {' '.join('%.3f' % t for t in ts[v])})")
