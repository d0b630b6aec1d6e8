"""Interleaved A/B of decode-step switches on the 7B engine (random embeddings prefilled, no ViT): each entry of --variants is a '+'-joined list of
trace_op_set_gemm_variant codes applied before that arm's steps (e.g. 180 / 181 = the SwiGLU fold off / on); eager launches, so a
switch takes effect at once; rounds interleaved, median; logits of the first steps compared across the arms.
python tools/decode_variant_ab.py --batch 128 --variants 180,181 [--ctx 1968] [--steps 24] [--reset 632]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="180,181")
ap.add_argument("--reset", default="0", help="codes applied at the end (the shipped defaults of the switches touched)")
a = ap.parse_args()
arms = [[int(x) for x in v.split("+")] for v in a.variants.split(",")]
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=a.batch, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for b in range(a.batch):
    eng.prefill(b, a.ctx, embeds=emb)
torch.cuda.synchronize()
slots = list(range(a.batch))
lg = []
for arm in arms:
    for v in arm: ops.set_gemm_variant(v)
    steps = [eng.decode_begin(slots, [1] * a.batch, 256, eos=-1, want_logits=True).clone()]
    for _ in range(3):
        steps.append(eng.decode_steps(1, use_graph=False, want_logits=True).clone())
    lg.append(torch.stack(steps))
print("logits of 4 steps identical across arms:", [bool(torch.equal(lg[0], x)) for x in lg])
ts = [[] for _ in arms]
for rnd in range(a.rounds + 1):
    for i, arm in enumerate(arms):
        for v in arm: ops.set_gemm_variant(v)
        eng.decode_begin(slots, [1] * a.batch, 256, eos=-1)        # every round restarts at the prefilled context
        eng.decode_steps(2, use_graph=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_steps(a.steps, use_graph=False)
        torch.cuda.synchronize()
        ts[i].append((time.perf_counter() - t0) / a.steps * 1e3)
for v in a.reset.split("+"):
    ops.set_gemm_variant(int(v))
for i, arm in enumerate(arms):
    m = statistics.median(ts[i][1:])
    print(f"batch {a.batch} ctx {a.ctx}: variants {'+'.join(map(str, arm)):12s} {m:.3f} ms/step = {a.batch * 1e3 / m:.0f} tok/s  (rounds: {' '.join('%.3f' % t for t in ts[i])})", flush=True)
eng.close()
