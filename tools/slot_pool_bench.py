"""SpatialSlotPool at the TRACE-7B geometry: 128 frames x 576 patches x 1024 -> 8 slots (slot_pool_part + merge kernels + readout GEMM)."""
import dataclasses, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine
cfg = dataclasses.replace(tcfg.trace_7b(128), num_hidden_layers=1, vision_num_layers=2)
eng = TraceEngine(cfg, max_batch=1, max_ctx=2304, max_frames=128, max_new_tokens=8)
eng.load_weights(synth.iter_weights(cfg, device="cuda:0"))
feats = (torch.randn(128, 576, 1024, device="cuda:0")).to(torch.bfloat16)
for _ in range(3): eng.slot_pool(feats, 128)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): eng.slot_pool(feats, 128)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
print(f"slot pool + readout, 128 frames: {us:.1f} us per call ({128 * 576 * 1024 * 2 / us / 1e6:.2f} TB/s of input)")
eng.close()
