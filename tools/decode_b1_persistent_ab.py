"""Batch-1 decode step (the reference drivers' call shape): the launch-per-kernel step (variant 900: 32 x 5 kernels + norm + heads under a hipGraph) against the
persistent single-launch step of decode_b1.hip (901 / 902 / 903: no / one / two weight load batches requested in front of each grid barrier).  One engine per arm,
each capturing its hipGraph under its own variant; interleaved rounds of graph replays, median; logits of 7 eager steps compared bit for bit; decode_read() raises
if a barrier timed out.   python tools/decode_b1_persistent_ab.py [--ctx 1968] [--arms 900,901,902,903]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops

ap = argparse.ArgumentParser()
ap.add_argument("--ctx", type=int, default=1968)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--arms", default="900,901,902,903")
ap.add_argument("--layers", type=int, default=0, help="decoder layers (0 = the 7B model's 32)")
a = ap.parse_args()
import dataclasses
cfg = tcfg.trace_7b()
if a.layers:
    cfg = dataclasses.replace(cfg, num_hidden_layers=a.layers)
names = {900: "launch per kernel", 901: "persistent, no loads before barriers", 902: "persistent, one load batch ahead", 903: "persistent, two load batches ahead",
         906: "persistent, one batch ahead, wave 0 polls first", 907: "persistent, two batches ahead, wave 0 polls first",
         917: "TIMING ONLY: 901 without acquire fences", 923: "TIMING ONLY: 907 without acquire fences", 965: "TIMING ONLY: 901 with free barriers",
         971: "TIMING ONLY: 907 with free barriers"}
arms = [int(x) for x in a.arms.split(",")]
lg = {}
torch.manual_seed(0)
emb = (torch.randn(a.ctx, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
# ONE engine per arm, created, measured and closed under its own variant (the switch is process-wide: an engine's decode_begin / graph capture follow the value
# it has at that moment); rounds are therefore not interleaved — run the tool twice to see the box's repeatability
ts = {v: [] for v in arms}
for v in arms:
    ops.set_gemm_variant(v)
    e = TraceEngine(cfg, max_batch=1, max_ctx=a.ctx + 320, max_frames=128, max_new_tokens=256)
    e.load_weights(synth.iter_weights(cfg, device="cuda"))
    e.prefill(0, a.ctx, embeds=emb)
    steps = [e.decode_begin([0], [1], 256, eos=-1, want_logits=True).clone()]
    for _ in range(6):
        steps.append(e.decode_steps(1, use_graph=False, want_logits=True).clone())
    e.decode_read()
    lg[v] = torch.stack(steps)
    print(f"arm {v}: logits over 7 eager steps bit-identical to arm {arms[0]}: {torch.equal(lg[v], lg[arms[0]])}", flush=True)
    for rnd in range(6):
        e.decode_begin([0], [1], 256, eos=-1)
        e.decode_steps(2, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.decode_steps(a.steps, use_graph=True)
        torch.cuda.synchronize()
        ts[v].append((time.perf_counter() - t0) / a.steps * 1e3)
        e.decode_read()
    e.close()
ops.set_gemm_variant(907)
for v in arms:
    m = statistics.median(ts[v][1:])
    print(f"batch 1 ctx {a.ctx}: {names.get(v, str(v)):48s} {m:.3f} ms/step = {1e3 / m:.0f} tok/s  (rounds: {' '.join('%.3f' % t for t in ts[v])})")
