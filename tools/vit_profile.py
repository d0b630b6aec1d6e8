"""ViT + slot-pool + prefill workload for rocprofv3 (TRACE-7B geometry, 128 frames, one video)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=1, max_ctx=2304, max_frames=128, max_new_tokens=256)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
frames = synth.synth_frames(cfg, 0).to(torch.bfloat16).cuda()
ts = [[float(i)] for i in range(128)]
ids = synth.synth_prompt_ids(cfg).tolist()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.encode_video(frames, ts); torch.cuda.synchronize(); t1 = time.perf_counter()
    L = eng.splice(ids); eng.prefill(0, L); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"encode_video {1e3*(t1-t0):.2f} ms   prefill(L={L}) {1e3*(t2-t1):.2f} ms")
