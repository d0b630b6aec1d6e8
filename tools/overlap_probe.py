"""Does encode (ViT + slot pool + prefill: MFMA-bound) overlap with decode (HBM-bound) when issued on two HIP streams — plain streams,
and CU-partitioned streams (trace_stream_create: decode confined to N CUs, encode to the rest, persistent GEMM grid capped)?

    python tools/overlap_probe.py [--B 64] [--nenc 32] [--modes 0,32,48,64,96] [--steps 255]

One JSON line per mode: decode alone / encode alone on that mode's streams, both together, and what a two-stage pipeline over batches of
B videos would deliver (videos/s = B / max-stage time when both stages run side by side, vs B / (dec + enc) back to back)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth  # noqa: E402
from trace_amd.engine import TraceEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--nenc", type=int, default=64, help="videos encoded + prefilled in the encode leg")
ap.add_argument("--modes", default="0,32,64,96")
ap.add_argument("--steps", type=int, default=255)
args = ap.parse_args()
B, NE, NS = args.B, args.nenc, args.steps

cfg = tcfg.trace_7b()
n_text = 176
L = n_text - 1 + 128 * cfg.tokens_per_frame
eng = TraceEngine(cfg, max_batch=2 * B, max_ctx=(L + NS + 1 + 63) // 64 * 64, max_frames=128, max_new_tokens=NS + 1)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
videos = [synth.synth_frames(cfg, b, dtype=torch.bfloat16, device="cuda") for b in range(4)]
videos = [videos[b % 4] for b in range(max(B, NE))]
ts = [[[float(i)] for i in range(128)]] * max(B, NE)
ids = [synth.synth_prompt_ids(cfg, n_text=n_text, video_pos=150).tolist()] * max(B, NE)
eng.encode_prefill(videos[:B], ts[:B], ids[:B], 0)             # bank 0: the batch that decodes
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0


for mode in [int(x) for x in args.modes.split(",")]:
    enc_s, dec_s = eng.make_streams(mode)

    def dec(graph=True):
        with torch.cuda.stream(dec_s):
            eng.decode_begin(list(range(B)), [1] * B, NS + 1)
            eng.decode_steps(NS, use_graph=graph)

    def enc():
        with torch.cuda.stream(enc_s):
            eng.encode_prefill(videos[:NE], ts[:NE], ids[:NE], B)

    dec(); torch.cuda.synchronize()
    t_dec = timed(dec)
    t_dec_eager = float("nan")
    enc(); torch.cuda.synchronize()
    t_enc = timed(enc)
    def both():
        # the decode leg from its own host thread: ~75 k dispatches do not fit the stream's queue, the issuing thread blocks until the GPU
        # has consumed them (issued from one thread the legs serialise: profiles/r03_overlap_probe_single_thread.jsonl)
        import threading
        th = threading.Thread(target=lambda: (torch.cuda.set_device(0), dec()))
        th.start(); enc(); th.join()
    t_both = timed(both)
    # per-batch stage times when run side by side: scale the encode leg to B videos
    seq = t_dec + t_enc * B / NE
    # in the joint run the two legs end at different times; the pipeline's steady state is bounded by the slower stage under contention.
    # Estimate it from the joint run: total work done = 1 decode batch + NE/B of an encode batch in t_both
    print(json.dumps({"mode": "plain streams" if mode == 0 else f"decode on {mode} CUs, encode on {256 - mode}", "B": B, "nenc": NE,
                      "decode_alone_ms": t_dec * 1e3, "decode_alone_eager_ms": t_dec_eager * 1e3, "ms_per_step": t_dec * 1e3 / NS,
                      "encode_alone_ms": t_enc * 1e3, "encode_ms_per_video": t_enc * 1e3 / NE, "both_ms": t_both * 1e3,
                      "sum_ms": (t_dec + t_enc) * 1e3, "overlap_gain": (t_dec + t_enc) / t_both, "pipelined_videos_s": B / t_both,
                      "back_to_back_videos_s": B / seq}), flush=True)
    eng.lib.trace_set_gemm_cus(eng.h, 0)
eng.close()
