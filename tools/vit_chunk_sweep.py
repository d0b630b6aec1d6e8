"""ViT tower time per 128-frame video as a function of the frames pushed through one trace_vit_forward call
(a stream of 32 videos = 4096 frames cut into chunks of F, remainder chunk included).  usage: python tools/vit_chunk_sweep.py [F ...]"""
import dataclasses, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine

cfg = dataclasses.replace(tcfg.trace_7b(128), num_hidden_layers=1)
Fs = [int(x) for x in sys.argv[1:]] or [64, 96, 113, 128, 142, 156, 170, 184, 198, 213, 227, 256]
eng = TraceEngine(cfg, max_batch=2, max_ctx=2304, max_frames=128, max_new_tokens=8, vit_batch_frames=max(Fs))
eng.load_weights(synth.iter_weights(cfg, device="cuda:0"))
nv = 32
vids = [synth.synth_frames(cfg, b, num_frames=128, dtype=torch.bfloat16, device="cuda:0") for b in range(nv)]
for F in Fs:
    eng.vit_batch_frames = F
    eng.vit_forward_many(vids[:4])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    eng.vit_forward_many(vids)
    b.record()
    torch.cuda.synchronize()
    rows = F * 577
    print("F=%3d  row tiles %6.1f  %.2f ms per 128-frame video" % (F, rows / 256.0, a.elapsed_time(b) / nv), flush=True)
eng.close()
