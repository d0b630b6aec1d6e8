"""Does encode (ViT + prefill, MFMA-bound) overlap with decode (HBM-bound) when issued on two HIP streams?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NENC = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=2 * B, max_ctx=2304, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
frames = synth.synth_frames(cfg, 0).to(torch.bfloat16).cuda()
ts = [[float(i)] for i in range(128)]
ids = synth.synth_prompt_ids(cfg).tolist()
emb = (torch.randn(1967, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
for b in range(B):
    eng.prefill(b, 1967, embeds=emb)
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def dec():
    with torch.cuda.stream(sA):
        eng.decode_begin(list(range(B)), [1] * B, 256)
        eng.decode_steps(255, use_graph=True)

def enc(n):
    with torch.cuda.stream(sB):
        for i in range(n):
            eng.encode_video(frames, ts)
            eng.prefill(B + (i % B), eng.splice(ids))

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0

dec(); enc(1); torch.cuda.synchronize()
ta = timed(dec)
tb = timed(lambda: enc(NENC))
tc = timed(lambda: (dec(), enc(NENC)))
print(f"B={B} decode alone {ta*1e3:.0f} ms | encode x{NENC} alone {tb*1e3:.0f} ms | both {tc*1e3:.0f} ms | sum {1e3*(ta+tb):.0f} ms | overlap gain {(ta+tb)/tc:.2f}x")
