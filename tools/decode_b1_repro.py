"""Repeatability of the batch-1 decode step under a variant: R repetitions of (decode_begin, N free-running steps) on one prefilled context; the ids of every repetition
and the logits of the last step must be identical (and equal to the launch-per-kernel step's, variant 900).  python tools/decode_b1_repro.py --variants 900,907,901 [--steps 96]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--variants", default="900,907")
ap.add_argument("--graph", type=int, default=1)
a = ap.parse_args()
cfg = tcfg.trace_7b()
torch.manual_seed(0)
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
ref = None
for v in [int(x) for x in a.variants.split(",")]:
    ops.set_gemm_variant(v)
    e = TraceEngine(cfg, max_batch=1, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
    e.load_weights(synth.iter_weights(cfg, device="cuda"))
    e.prefill(0, a.ctx, embeds=emb)
    runs = []
    for r in range(a.reps):
        e.decode_begin([0], [1], 256, eos=-1)
        e.decode_steps(a.steps - 1, use_graph=bool(a.graph))
        lg = e.decode_steps(1, use_graph=False, want_logits=True).float().cpu()
        runs.append((e.decode_read()[0][0], lg))
    same = [runs[i][0] == runs[0][0] and torch.equal(runs[i][1], runs[0][1]) for i in range(a.reps)]
    first = [next((k for k, (x, y) in enumerate(zip(runs[i][0], runs[0][0])) if x != y), None) for i in range(a.reps)]
    if ref is None:
        ref = runs[0]
    vs_ref = runs[0][0] == ref[0] and torch.equal(runs[0][1], ref[1])
    print(f"variant {v}: repetitions identical {same} (first differing token {first}); equal to variant {a.variants.split(',')[0]}: {vs_ref}", flush=True)
    e.close()
ops.set_gemm_variant(907)
