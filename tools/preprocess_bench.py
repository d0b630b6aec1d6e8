"""Frame preprocessing: GPU (trace_preprocess_frames) vs the host path the reference uses (PIL + HF CLIPImageProcessor),
128 frames of 1280x720 -> [128,3,336,336]."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, mm_utils
from trace_amd.engine import TraceEngine
import dataclasses
cfg = dataclasses.replace(tcfg.tiny(num_frames=128), vision_image_size=336, vision_patch_size=14)
eng = TraceEngine(cfg, max_batch=1, max_ctx=4096, max_frames=128, max_new_tokens=8)
raw = np.random.RandomState(0).randint(0, 256, size=(128, 720, 1280, 3), dtype=np.uint8)
dev = torch.from_numpy(raw).cuda()
for pad in (True, False):
    eng.preprocess_frames(dev, pad=pad); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): out = eng.preprocess_frames(dev, pad=pad)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    t0 = time.perf_counter(); out = eng.preprocess_frames(raw, pad=pad); torch.cuda.synchronize(); h2d = (time.perf_counter() - t0) * 1e3
    print(f"GPU pad={pad}: {ms:.2f} ms for 128 frames 1280x720 (device-resident uint8; {raw.nbytes / ms / 1e6:.0f} GB/s of input); "
          f"{h2d:.1f} ms including the 354 MB host->device copy", flush=True)
from transformers import CLIPImageProcessor
proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
t0 = time.perf_counter()
host, _ = mm_utils.process_video(raw[:16], proc, aspect_ratio="pad", num_frames=16, fps=1.0)
t = (time.perf_counter() - t0) * 8
print(f"host PIL/HF path: {t * 1e3:.0f} ms per 128 frames (16 measured x 8), one core")
assert torch.equal(eng.preprocess_frames(raw[:16], pad=True).cpu(), host.to(torch.bfloat16))
print("device == host path (bf16 of it): ok")
