#!/usr/bin/env python3
"""Headline benchmark: end-to-end TRACE-7B video-grounding inference on synthetic 128-frame x 336^2 clips.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5]
        N > 1 without WORLD_SIZE in the environment: bench.py launches its own ranks (python -m torch.distributed.run
        --nproc-per-node N, rendezvous on 127.0.0.1) and fails if fewer than N GPUs are visible; under an external
        torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE and requires WORLD_SIZE == N.

A "step" = one pass of the hot path over one batch of `--videos-per-step` (default 128; 32 for c5) videos per GPU, inputs already
resident in HBM: CLIP-ViT-L/14-336 over 128 frames -> SpatialSlotPool -> splice -> Mistral-7B prefill (L = 1967) -> 256 greedy
decode steps with head switching, then ONE RCCL all-gather of the packed token ids (N > 1).  Prints one JSON line
(rank 0) with the whole-job videos/sec, the decode tokens/sec, `roofline` = the dominant kernel family of the run (the 256x256
MFMA GEMM: gemm_ldr_kernel<1> is the top symbol by total time, gemm_w4_kernel<2> its largest single launch; all four ViT shapes are
bracketed with HIP events inside the timed region), `roofline_hbm` = the dominant HBM-bound kernel of the decode phase (the decode
attention at the default batch), per-rank timings and host placement (N > 1), and the CPU baseline (the oracle timed on a bounded sample
on this box's host cores; N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from trace_amd import config as tcfg, dist as tdist, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(cfg, n_new: int, threads: int) -> dict:
    """Oracle (CPU restatement of the reference path, validated against the reference fixtures) timed on a bounded
    sample of the same workload and extrapolated linearly: ViT on 8 of the 128 frames (all 23 layers; SURVEY 8d's sample), slot pool on
    those frames, 4 of the 32 decoder layers at the full prefill length, and 8 single-token decode steps on those layers."""
    import dataclasses
    from oracle import trace_oracle as O
    torch.set_num_threads(threads)
    SF, SL, SD = 8, 4, 8                      # sample: frames, decoder layers, decode steps (~12 s of work on 32 threads)
    c1 = dataclasses.replace(cfg, num_hidden_layers=SL, vocab_size=64, num_frames=SF)
    sd = {}
    for name, shape, kind in synth.weight_specs(c1):
        sd[name] = synth.synth_tensor(name, shape, kind, torch.float32)
    ora = O.Oracle(c1, sd, emulate_bf16=False)
    T_full, L_full = cfg.num_frames, cfg.num_frames * cfg.tokens_per_frame + 176
    frames = synth.synth_frames(c1, 0, num_frames=SF)
    with torch.no_grad():
        t0 = time.perf_counter(); feats = ora.vit_forward(frames); t_vit = time.perf_counter() - t0
        t0 = time.perf_counter(); ora.slot_pool(feats); t_slot = time.perf_counter() - t0
        emb = torch.randn(L_full, cfg.hidden_size) * 0.02
        t0 = time.perf_counter(); _, kv = ora.llm_forward(emb); t_pre = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(SD):
            _, kv = ora.llm_forward(torch.randn(1, cfg.hidden_size) * 0.02, kv)
        t_dec = (time.perf_counter() - t0) / SD
    NLf = cfg.num_hidden_layers
    per_video = (t_vit + t_slot) * T_full / SF + t_pre * NLf / SL + t_dec * NLf / SL * (n_new - 1)
    return {"value": 1.0 / per_video, "unit": "videos/s", "cores": threads, "kind": "port",
            "sample": (f"oracle fp32: ViT+slot-pool on {SF}/{T_full} frames ({t_vit + t_slot:.2f}s), {SL}/{NLf} decoder layers at L={L_full} "
                       f"({t_pre:.2f}s), {SD} decode steps on those layers ({t_dec * 1e3:.1f} ms each); extrapolated linearly "
                       f"(heads/embedding excluded) -> {per_video:.0f} s/video"),
            "decode_tok_s": 1.0 / (t_dec * NLf / SL)}


def dvc_schedule(cfg, n_new, seed):
    """Forced feed ids giving every sequence the dense-video-captioning head pattern of the reference's outputs
    (SURVEY.md section 8d): per event 14 time-head steps ("0012.3<sep>0045.6" + time <sync>), 4 score-head steps ("4.5" +
    score <sync>) and 33 text-head steps (32 caption tokens + text <sync>), i.e. the 262 MB text head is streamed on
    ~65 % of the steps as in real use.  The engine still computes the active head's logits and arg-max every step
    (trace_mistral.py:244-252); only the token fed back is fixed, so the head state machine walks this schedule."""
    rng = np.random.RandomState(1000 + seed)
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    out = []
    while len(out) < n_new:
        digits = lambda n, base: [base + 3 + int(d) for d in rng.randint(0, 10, size=n)]    # '0'..'9' = base+3 .. base+12
        t = digits(4, V) + [V + 13] + digits(1, V)                  # dddd.d
        out += t + [V + 2] + digits(4, V) + [V + 13] + digits(1, V) + [V + 1]       # <sep> ... time <sync> -> score head
        sb = V + Tv
        out += digits(1, sb) + [sb + 13] + digits(1, sb) + [sb + 1]                 # d.d score <sync> -> text head
        out += [int(x) for x in rng.randint(3, V, size=32)] + [V]                   # caption, text <sync> -> time head
    return out[:n_new]


def mr_schedule(cfg, n_new, seed):
    """Forced feed for the moment-retrieval answer shape of BASELINE config 4 (trace/eval/evaluate.py:298-357 with prompts/mr.txt):
    one event = 14 time-head steps, 4 score-head steps, then caption text and the text <sync> (18 of 32 steps on the 13-way heads)."""
    rng = np.random.RandomState(2000 + seed)
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    digits = lambda n, base: [base + 3 + int(d) for d in rng.randint(0, 10, size=n)]
    sb = V + Tv
    out = []
    while len(out) < n_new:
        out += digits(4, V) + [V + 13] + digits(1, V) + [V + 2] + digits(4, V) + [V + 13] + digits(1, V) + [V + 1]
        out += digits(1, sb) + [sb + 13] + digits(1, sb) + [sb + 1]
        out += [int(x) for x in rng.randint(3, V, size=13)] + [V]
    return out[:n_new]


# BASELINE.json configs (C1 = the reference's CPU-runnable plumbing case: run_c1 below; C3 = C2 under --gpus 8)
CONFIGS = {
    "c1": dict(frames=8, max_new=32, n_text=0, video_pos=0, schedule="none", videos_per_step=1,
               name="C1: single clip from a file, 8 frames, greedy decode through the drivers' call sequence (scripts/inference/inference.py), tiny-layer model"),
    "c2": dict(frames=128, max_new=256, n_text=176, video_pos=150, schedule="dvc", videos_per_step=128,
               name="C2: TRACE-7B bf16 (CLIP-ViT-L/14-336 23 layers + SpatialSlotPool + Mistral-7B)"),
    "c4": dict(frames=64, max_new=32, n_text=191, video_pos=150, schedule="mr", videos_per_step=128,
               name="C4: Charades-STA moment retrieval shape, TRACE-7B bf16"),
    "c5": dict(frames=256, max_new=16, n_text=251, video_pos=200, schedule="dvc", videos_per_step=32, fp8=True,
               name="C5: VideoMME long video (256 frames, past MAX_FRAMES), TRACE-7B, fp8 (e4m3 W8A8) decoder projections"),
    "stc": dict(frames=16, max_new=0, n_text=0, video_pos=0, schedule="none", videos_per_step=1,
                name="STC connector (legacy trace.infer path) at its real shape: 16 frames x 24x24 patches, 1024 -> 4096 channels -> 9 x 13 x 13 tokens"),
}


def synthetic_clip(T=96, H=120, W=160):
    """a deterministic moving-pattern clip (uint8 RGB [T, H, W, 3]): stands in for the reference's assets/sora.mp4, which is a missing blob"""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    t = np.arange(T)[:, None, None]
    r = (xx[None] * 2 + t * 5) % 256
    g = (yy[None] * 3 + t * 3) % 256
    b = ((xx[None] + yy[None]) + t * 7) % 256
    box = ((xx[None] - (20 + t)) % W < 24) & ((yy[None] - 30) % H < 24)          # a square drifting to the right
    out = np.stack([r, g, b], -1).astype(np.uint8)
    out[np.broadcast_to(box, out.shape[:3])] = 255
    return out


def run_c1(args) -> None:
    """BASELINE config 1 — "single clip, 8 frames, greedy decode on CPU via scripts/inference/inference.py (plumbing)": the drivers' call sequence
    (inference.py:32-128: process_video from a FILE -> conversation prompt + <sync> -> tokenizer_MMODAL_token_all -> generate(heads=[1]) -> id parser)
    on a synthetic clip written as YUV4MPEG2 (no decord here; trace_amd/video_io.py), T = 8, 32 greedy tokens, the tiny-layer model of the reference
    fixtures (hidden 4096 is forced by trace_arch.py:38-40).  Two legs on the same inputs:
      * CPU: the oracle (the reference path restated, pinned by the reference fixtures) end to end — BASELINE.md section 3's "C1 runs fully on CPU";
      * GPU (when one is visible): load_pretrained_model -> process_video(file) -> model.generate through the C ABI; ids compared with the oracle's.
    `value` = the GPU leg's videos/s (file read + preprocessing + generate per step); without a GPU the line reports the CPU leg only."""
    import tempfile
    from oracle import trace_oracle as O                      # checker + CPU baseline only
    from trace_amd import video_io
    from trace_amd.constants import DEFAULT_MMODAL_TOKEN
    from trace_amd.conversation import conv_templates
    from trace_amd.evaluate import parse_output_ids
    from trace_amd.mm_utils import get_model_name_from_path, process_video, tokenizer_MMODAL_token_all
    from trace_amd.model import builder
    cfg = tcfg.tiny(num_frames=args.frames)
    n_new = args.max_new
    tmp = tempfile.mkdtemp(prefix="trace_c1_")
    clip = os.path.join(tmp, "clip.y4m")
    video_io.write_y4m(clip, synthetic_clip(), fps=(24, 1), chroma="420")
    ckpt = builder.save_synthetic_checkpoint(os.path.join(tmp, "trace-tiny"), cfg)
    question = "Find the events of this clip: give the start and end time of each, a score, and one sentence about it."

    def driver_inputs(tokenizer, processor, engine=None):
        tensor, ts = process_video(clip, processor, "pad", num_frames=args.frames, engine=engine)        # inference.py:34
        conv = conv_templates["llama_2"].copy()                                                           # inference.py:50-55
        conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\n" + question)
        conv.append_message(conv.roles[1], None)
        prompt = conv.get_prompt() + "<sync>"
        return tensor, ts, tokenizer_MMODAL_token_all(prompt, tokenizer, return_tensors="pt")

    # ---- CPU leg: the oracle end to end on this box's host cores (no extrapolation)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(32, cores))
    torch.set_num_threads(threads)
    tok_cpu, proc_cpu = builder.ByteTokenizer(cfg.vocab_size), builder._image_processor(cfg, ckpt)
    sd = synth.state_dict(cfg)
    t0 = time.perf_counter()
    tensor, ts, ids = driver_inputs(tok_cpu, proc_cpu)
    t_pre_cpu = time.perf_counter() - t0
    ora = O.Oracle(cfg, {k: v.float() for k, v in sd.items()}, emulate_bf16=False)
    with torch.no_grad():
        t0 = time.perf_counter()
        ids_fp32 = ora.generate(ids, tensor.float(), ts, head=1, max_new_tokens=n_new, eos_token_id=None)
        t_gen_cpu = time.perf_counter() - t0
    cpu = {"value": 1.0 / (t_pre_cpu + t_gen_cpu), "unit": "videos/s", "cores": threads, "kind": "port",
           "sample": f"the whole C1 workload, no extrapolation: file read + PIL/HF preprocessing {t_pre_cpu * 1e3:.0f} ms, oracle fp32 ViT + slot pool + prefill L={len(ids) - 1 + args.frames * cfg.tokens_per_frame} + {n_new} greedy tokens {t_gen_cpu * 1e3:.0f} ms",
           "decode_tok_s": None, "ids": ids_fp32}
    line = {"metric": "videos/sec + decode tok/s, TRACE-7B 128-frame, 1/2/4/8 MI355X", "value": None, "unit": "videos/s", "n_gpus": 0, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{CONFIGS['c1']['name']}: {args.frames} frames sampled from a 96-frame 160x120 y4m file, prefill L={len(ids) - 1 + args.frames * cfg.tokens_per_frame}, {n_new} greedy tokens, heads=[1], EOS off",
                       "baseline_config": "c1", "videos_per_step_per_gpu": 1, "frames": args.frames, "new_tokens": n_new,
                       "model": "tiny-layer geometry of the reference fixtures (hidden 4096, 2 decoder layers, 3-layer CLIP at 56x56), synthetic weights"},
            "cpu_baseline": cpu}
    import shutil
    if not torch.cuda.is_available():
        line["config"]["note"] = "no HIP device visible: CPU leg only (the HIP path has no CPU fallback)"
        print(json.dumps(line), flush=True)
        shutil.rmtree(tmp, ignore_errors=True)
        return
    # ---- GPU leg: the drop-in surface over the C ABI
    el_dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    tokenizer, model, processor, _ = builder.load_pretrained_model(ckpt, None, get_model_name_from_path(ckpt), max_new_tokens=max(64, n_new), torch_dtype=el_dtype)

    def step():
        tensor_g, ts_g, ids_g = driver_inputs(tokenizer, processor, engine=model)          # frames preprocessed on the device (trace_preprocess_frames)
        heads = [1]
        out = model.generate(ids_g.unsqueeze(0), attention_mask=None, images_or_videos=[tensor_g], modal_list=["video"], do_sample=False, temperature=0.0,
                             max_new_tokens=n_new, use_cache=True, pad_token_id=tokenizer.eos_token_id, eos_token_id=-1, video_timestamps=[ts_g], heads=heads)
        return out[0].tolist(), ts_g, tensor_g
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids_hip, ts_g, tensor_g = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # ids: against the oracle that rounds where the engine stores 16-bit values, on the frames the device preprocessing produced
    ora_b = O.Oracle(cfg, sd, emulate_bf16=(True if el_dtype == torch.bfloat16 else el_dtype))
    with torch.no_grad():
        ids_emu, lg = ora_b.generate(ids, tensor_g.float().cpu(), ts_g, head=1, max_new_tokens=n_new, eos_token_id=None, return_logits=True)
    srt = torch.sort(torch.where(torch.isfinite(lg), lg, torch.full_like(lg, -1e30)), dim=-1, descending=True).values
    margin = (srt[:, 0] - srt[:, 1]).tolist()
    first_diff = next((i for i, (a, b) in enumerate(zip(ids_hip, ids_emu)) if a != b), None)
    V = cfg.vocab_size
    line.update({"value": args.steps / dt, "n_gpus": 1, "ms_per_step": dt / args.steps * 1e3,
                 "ids": {"hip": ids_hip, "oracle_16bit_emulating": ids_emu, "oracle_fp32_host_preprocessing": ids_fp32, "equal": ids_hip == ids_emu,
                         "first_difference": first_diff, "oracle_top2_margin_there": (margin[first_diff] if first_diff is not None else None),
                         "time_score_head_steps": sum(1 for t in ids_emu if t > V)},
                 "parsed": parse_output_ids(ids_hip, tokenizer, model)})
    print(json.dumps(line), flush=True)
    model.engine.close()
    shutil.rmtree(tmp, ignore_errors=True)


def stc_flops(T: int, G: int, cin: int, H: int) -> dict:
    """algorithmic FLOPs of one STCConnector.forward (projector/builder.py:208-249) by part: the 1x1 convolutions of the two RegStages (timm Bottleneck:
    conv1, conv3 and, in a stage's first block when the width changes, the shortcut conv), the Conv3d sampler as a GEMM, the readout MLP; the depthwise
    3x3 convolutions and the squeeze-excite GEMVs are counted too (VALU / GEMV work, not MFMA)."""
    rows1, To, Go = T * G * G, T // 2 + 1, G // 2 + 1
    rows2 = To * Go * Go
    s1 = 2.0 * rows1 * H * cin * 2 + 2.0 * rows1 * H * H + 3 * 2 * 2.0 * rows1 * H * H      # block 0: conv1 + shortcut (cin -> H), conv3; blocks 1-3: conv1, conv3
    s2 = 4 * 2 * 2.0 * rows2 * H * H
    return {"s1_convs": s1, "sampler": 2.0 * rows2 * H * 8 * H, "s2_convs": s2, "readout": 2 * 2.0 * rows2 * H * H,
            "depthwise": 2.0 * 9 * H * (4 * rows1 + 4 * rows2), "squeeze_excite": 8 * 2 * 2.0 * max(T, To) * H * (H // 4), "rows1": rows1, "rows2": rows2}


def run_stc(args) -> None:
    """`--config stc`: the STC connector at the shape the reference builds it with (mm_hidden 1024 -> hidden 4096, 24 x 24 patch grid, 16
    frames -> (16 / 2 + 1) x 13 x 13 = 1521 tokens), on patch features resident in HBM: ms per call, TFLOP/s and the fraction of the bf16 MFMA peak over the
    connector's GEMM work.  Weights are synthetic (device RNG); the RegStage block is the restated one (timm is not importable offline: DESIGN section 2)."""
    import dataclasses
    from trace_amd.engine import TraceEngine
    if not torch.cuda.is_available():
        raise SystemExit("--config stc needs an MI355X (the HIP path has no CPU fallback)")
    T = args.frames
    cfg = dataclasses.replace(tcfg.trace_7b(T), mm_projector_type="stc_connector", num_hidden_layers=1, vision_num_layers=2)    # the LLM and the tower are not run here
    eng = TraceEngine(cfg, device=0, max_batch=1, max_ctx=2048, max_frames=T, max_new_tokens=8)
    eng.load_weights(synth.iter_weights(cfg, device="cuda"))
    feats = (torch.randn(T, cfg.vision_patches, cfg.vision_hidden_size, device="cuda") * 1.0).to(torch.bfloat16)
    for _ in range(max(1, args.warmup)):
        out = eng.stc_connector(feats, T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        out = eng.stc_connector(feats, T)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    fl = stc_flops(T, cfg.vision_grid, cfg.mm_hidden_size, cfg.hidden_size)
    gemm_fl = fl["s1_convs"] + fl["sampler"] + fl["s2_convs"] + fl["readout"]
    tf = gemm_fl / (ms * 1e-3) / 1e12
    print(json.dumps({
        "metric": "STC connector, ms per 16-frame clip (auxiliary line: SURVEY row a11)", "value": ms, "unit": "ms/call", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": CONFIGS["stc"]["name"], "frames": T, "tokens_out": int(out.shape[0]), "rows_stage1": fl["rows1"], "rows_stage2": fl["rows2"],
                   "weights": "random-init (device RNG); RegStage block restated (timm unavailable offline)"},
        "roofline": {"bound": "mfma", "kernel": "the connector's GEMMs together (1x1 convolutions of both RegStages, Conv3d sampler as an im2col GEMM, readout MLP) over the WHOLE call, elementwise passes included in the time",
                     "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None,
                     "algorithmic_gflop_per_call": gemm_fl / 1e9,
                     "gflop_by_part": {k: v / 1e9 for k, v in fl.items() if k not in ("rows1", "rows2")}},
        "finite": bool(torch.isfinite(out).all())}), flush=True)
    eng.close()


def self_launch(args) -> None:
    """`python bench.py --gpus N` with no launcher around it: start one rank per GPU ourselves (the driver's own N > 1 invocation,
    `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, sets WORLD_SIZE and never gets here)."""
    import socket
    import subprocess
    if not args.plumbing_check:
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible on this node — refusing to report a "
                             f"{args.gpus}-GPU number from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def plumbing_check(args, rank, world) -> None:
    """Launcher / rendezvous / gather path with no kernels (runs on CPU over gloo: tests/test_bench_launcher.py): every rank fabricates
    its videos' ids, the packed all-gather runs, rank 0 prints a line whose n_gpus / rccl_ranks prove every rank took part."""
    B, n_new = 3, 8
    place = tdist.bind_rank(int(os.environ.get("LOCAL_RANK", rank)), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    local = [[(1000 * rank + 10 * b + i) % 32027 for i in range(1 + (b + rank) % n_new)] for b in range(B)]
    tdist.barrier()
    t0 = time.perf_counter()
    g = tdist.gather_outputs(local, n_new, B, None if torch.cuda.is_available() else torch.device("cpu"))
    dt_local = time.perf_counter() - t0
    tdist.barrier()
    ok = all(g[r][b] == [(1000 * r + 10 * b + i) % 32027 for i in range(1 + (b + r) % n_new)] for r in range(world) for b in range(B))
    per_rank = tdist.gather_floats([dt_local * 1e3, float(rank), float(len(os.sched_getaffinity(0)))])          # the real line's per-rank collective
    if rank == 0:
        print(json.dumps({"metric": "plumbing check (no kernels)", "value": 0.0, "unit": "videos/s", "n_gpus": world, "rccl_ranks": len(g),
                          "gather_ok": ok, "data": "none", "config": {"workload": "launcher + rendezvous + packed-id all-gather only"},
                          "per_rank": {"ms_per_step": [r[0] for r in per_rank], "rank": [int(r[1]) for r in per_rank], "cpus": [int(r[2]) for r in per_rank],
                                       "host_placement_rank0": place}}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json configuration (c3 = c2 with --gpus 8)")
    ap.add_argument("--videos-per-step", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--max-new", type=int, default=None)
    ap.add_argument("--fp8", dest="fp8", action="store_true", default=None, help="decoder projections on the fp8 (e4m3 W8A8) path (default: on for --config c5)")
    ap.add_argument("--no-fp8", dest="fp8", action="store_false")
    ap.add_argument("--fp8-scheme", choices=["w8a8", "weight_only"], default="weight_only",
                    help="fp8 path: W8A8 everywhere, or W8A8 prefill GEMMs + weight-only (bf16 activations) decode GEMVs (default; tests/test_gpu_fp8.py: fewer 13-way arg-max flips)")
    ap.add_argument("--plumbing-check", action="store_true", help="launcher / rendezvous / gather only, no kernels (CPU-testable)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=False,
                    help="replay the captured hipGraph for decode steps; the roofline_hbm bracket (a per-launch HIP-event pair, which a graph cannot carry) is "
                         "then taken in a short eager pass after the timed region and the line says so")
    ap.add_argument("--eager", dest="graph", action="store_false", help="decode steps as eager launches (the per-launch roofline probe sits in the timed region)")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true", default=True,
                    help="(the default since round 4) two-stage pipeline over the timed steps — batch k decodes on one stream while batch k+1 runs its ViT + prefill "
                         "on another (TraceEngine.generate_stream): +2-3 %% videos/s.  Round 3 kept it off: over ~100 pipelined steps one video's ViT features "
                         "did not repeat step 0 a few times; rounds 3-4 traced that to the configuration with the ViT's LayerNorm fold on (8 of 866 steps with it, 0 of 392 "
                         "without: profiles/r04_pipeline_stress_*.txt); the fold was never root-caused and left the product in round 5")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="timed steps strictly one after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16",
                    help="element type: bf16 (BASELINE's configs; libtrace_hip.so) or fp16 (the reference's own inference dtype; libtrace_hip_f16.so)")
    ap.add_argument("--tiny", action="store_true", help="tiny geometry (plumbing check)")
    ap.add_argument("--vit-batch", type=int, default=None, help="frames per ViT call (default: TraceEngine.full_round_frames)")
    args = ap.parse_args()
    preset = CONFIGS[args.config]
    if args.frames is None:
        args.frames = preset["frames"]
    if args.max_new is None:
        args.max_new = preset["max_new"]
    if args.videos_per_step is None:
        args.videos_per_step = int(os.environ.get("TRACE_BENCH_BATCH", preset["videos_per_step"]))
    if args.fp8 is None:
        args.fp8 = bool(preset.get("fp8", False))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.config == "c1":
        if args.gpus != 1:
            raise SystemExit("--config c1 is the single-clip plumbing case: one process, at most one GPU")
        return run_c1(args)
    if args.config == "stc":
        if args.gpus != 1:
            raise SystemExit("--config stc times one connector call shape on one GPU")
        return run_stc(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                       # never returns

    rank, local, world = tdist.init_from_env(None if torch.cuda.is_available() else "gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report the wrong device count")
    if args.plumbing_check:
        return plumbing_check(args, rank, world)
    if torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # host placement: each rank on the CPUs of its GPU's NUMA node (or an even slice of the allowed CPUs when the platform does not say): eight ranks
    # issue ~75 k launches per decode batch each, from two threads — they must not share cores
    place = tdist.bind_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    from trace_amd.engine import TraceEngine, ops

    cfg = tcfg.tiny(args.frames) if args.tiny else tcfg.trace_7b(args.frames)
    B, n_new = args.videos_per_step, args.max_new
    n_text = 24 if args.tiny else preset["n_text"]
    ids = synth.synth_prompt_ids(cfg, n_text=n_text, video_pos=10 if args.tiny else preset["video_pos"]).tolist()
    L = n_text - 1 + args.frames * cfg.tokens_per_frame
    pipelined = args.pipeline and 2 * B <= 512 and args.steps > 1          # two banks of KV slots (the engine holds at most 512)
    el_dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    if args.fp8 and args.dtype != "bf16":
        sys.exit("bench: the fp8 weight path exists in the bf16 library only (--config c5 --dtype fp16 needs --no-fp8)")
    eng = TraceEngine(cfg, device=local, max_batch=2 * B if pipelined else B, max_ctx=(L + n_new + 63) // 64 * 64, max_frames=args.frames,
                      max_new_tokens=n_new, vit_batch_frames=args.vit_batch or TraceEngine.full_round_frames(cfg), llm_fp8=(args.fp8_scheme if args.fp8 else False),
                      dtype=el_dtype)
    # the route of the GEMM shapes without a residual is SET here (gemm_w4.hip unless TRACE_GEMM_W4 says 0, read with C's atoi like the library would),
    # so the kernel symbol printed in the line is the one this process asked for rather than a second reading of the environment
    import re as _re
    _w4 = os.environ.get("TRACE_GEMM_W4")
    _m = _re.match(r"\s*[+-]?\d+", _w4) if _w4 is not None else None
    use_w4 = True if _w4 is None else (int(_m.group()) != 0 if _m else False)
    ops.use("f16" if args.dtype == "fp16" else "bf16")
    ops.set_gemm_variant(530 + int(use_w4))
    t0 = time.perf_counter()
    eng.load_weights(synth.iter_weights(cfg, dtype=el_dtype, device=str(dev)))
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    # inputs resident in HBM before the timed region
    videos = [synth.synth_frames(cfg, rank * B + b, num_frames=args.frames, dtype=el_dtype, device=dev) for b in range(B)]
    ts = [[[float(i)] for i in range(args.frames)] for _ in range(B)]
    prompt = [ids] * B
    heads = [1] * B
    sched = mr_schedule if preset["schedule"] == "mr" else dvc_schedule
    forced = [sched(cfg, n_new, seed=rank * B + b) for b in range(B)]
    ranks_seen = [1]

    def step():
        out, _ = eng.generate(videos, ts, prompt, heads, n_new, eos=-1, use_graph=args.graph, forced=forced)
        if world > 1 or torch.distributed.is_initialized():
            g = tdist.gather_outputs(out, n_new, B, dev)
            ranks_seen[0] = len(g)
            if len(g) != world or any(len(x) != B for x in g):
                raise SystemExit(f"bench: the all-gather returned {len(g)} ranks, expected {world}")
        return out

    def gather(out):
        if world > 1 or torch.distributed.is_initialized():
            g = tdist.gather_outputs(out, n_new, B, dev)
            ranks_seen[0] = len(g)
            if len(g) != world or any(len(x) != B for x in g):
                raise SystemExit(f"bench: the all-gather returned {len(g)} ranks, expected {world}")
        return out

    def run(k):
        """k steps: strictly one after the other, or as a two-stage pipeline over the steps (every step's work, the pipeline's fill and drain
        included, is inside the caller's timed region; the ids of step i are gathered as soon as its decode has finished)"""
        if not pipelined or k < 2:
            return [step() for _ in range(k)]
        batch = (videos, ts, prompt, heads, forced)
        return [gather(o[0]) for o in eng.generate_stream([batch] * k, n_new, eos=-1, use_graph=args.graph)]

    run(args.warmup)
    eng.set_profile(2)
    eng.stage_timing(True)               # event pairs around the tower / slot-pool / prefill calls of the batch(es) that have the GPU to themselves
    tdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run(args.steps)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0                 # this rank's own K steps (before the closing barrier): the per-rank figure of the line
    tdist.barrier(); torch.cuda.synchronize()
    dt = tdist.max_over_ranks(time.perf_counter() - t0)
    out = outs[-1]
    # every step runs the same inputs through a path whose reductions all have a fixed order: the ids must repeat exactly
    # (checked on every rank; reported in the line as `steps_repeat_exactly` — the run is not aborted: under torchrun one rank leaving would hang the
    # others in the closing collectives, and a throughput measured on a run with one differing sequence is still a throughput; the tests are where
    # a difference fails)
    repeat_detail = []
    if any(o != outs[0] for o in outs[1:]):
        for k, o in enumerate(outs[1:], 1):
            bad = [(b, next((i for i, (x, y) in enumerate(zip(o[b], outs[0][b])) if x != y), -1)) for b in range(len(o)) if o[b] != outs[0][b]]
            if bad:
                repeat_detail.append(f"step {k}: {len(bad)} of {len(o)} sequences differ, first (sequence, token) pairs {bad[:6]}")
        print(f"bench (rank {rank}): output ids differ between identical steps (a race or an unordered reduction)\n  " + "\n  ".join(repeat_detail[:12]),
              file=sys.stderr, flush=True)
    steps_repeat = tdist.max_over_ranks(1.0 if repeat_detail else 0.0) == 0.0          # over all ranks (a collective: every rank gets here)
    prof = eng.get_profile()
    in_region = eng.stage_times()        # the fill batch of the pipeline (or every batch of a sequential run): tower / slot pool / prefill ms by events
    eng.stage_timing(False)
    eng.set_profile(0)

    # stage breakdown (outside the timed region; rank 0 only prints it).  Every stage is run once untimed in the call shape it is then timed in
    # (round 5's single cold shot gave 0.367 / 0.400 for the same binary: profiles/README.md), then timed REPS times: median, min, max.
    REPS = 3

    def ev_stat(fn, reps=REPS):
        fn(); torch.cuda.synchronize()
        ts_ = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts_.append(a.elapsed_time(b))
        return float(np.median(ts_)), float(min(ts_)), float(max(ts_))
    if B >= 2 and eng.vit_batch_frames > args.frames:          # as in generate(): the tower runs over the batch's frame stream, 12 GB of features at a time
        def enc_all():
            for g0, g1 in eng.feature_groups(videos):
                for f, t in zip(eng.vit_forward_many(videos[g0:g1]), ts[g0:g1]):
                    eng.encode_features(f, t)
        enc3 = [x / B for x in ev_stat(enc_all)]
    else:
        enc3 = list(ev_stat(lambda: eng.encode_video(videos[0], ts[0])))
    t_enc = enc3[0]
    # the KV slots of the last timed batch are still prefilled (a decode only appends behind the prompt): the decode stage is re-timed on them; the
    # prefill stage is timed into slots of the other bank (pipelined) or over the first slots (the same prompts again)
    slot_base = ((args.steps - 1) % 2) * B if pipelined else 0
    pf_slot = (B if slot_base == 0 else 0) if pipelined else 0
    npf = min(B, eng.prefill_group(L))           # what encode_prefill() puts through one pass at this prompt length
    embs = []
    for b in range(npf):
        eng.encode_video(videos[b], ts[b])
        Ls, e = eng.splice(ids, want_output=True)
        embs.append(e.clone())
    if npf >= 3:                                     # as in generate(): runs of equal-length neighbours share one prefill pass
        pre3 = [x / npf for x in ev_stat(lambda: eng.prefill_multi(pf_slot, embs))]
    elif npf == 2:
        pre3 = [x / 2 for x in ev_stat(lambda: eng.prefill_pair(pf_slot, embs[0], embs[1]))]
    else:
        pre3 = list(ev_stat(lambda: eng.prefill(pf_slot, Ls, embeds=embs[0])))
    t_pre = pre3[0]

    def dec_all():
        eng.decode_begin(list(range(slot_base, slot_base + B)), heads, n_new, -1, forced)       # every repetition restarts at the prefilled context
        eng.decode_steps(n_new - 1, use_graph=args.graph)
    dec3 = list(ev_stat(dec_all, reps=2 if n_new > 64 else REPS))
    t_dec = dec3[0]
    prof_k = prof
    if args.graph and n_new > 1:
        # hipGraph replay leaves no room for per-launch event brackets (event-record nodes inside a graph give no usable timestamps on ROCm 7.2):
        # the `roofline_hbm` bracket is taken here, in a short EAGER pass over the same prefilled slots, outside the timed region
        eng.set_profile(2)
        eng.decode_begin(list(range(slot_base, slot_base + B)), heads, n_new, -1, forced)
        eng.decode_steps(min(n_new - 1, 32), use_graph=False)
        torch.cuda.synchronize()
        prof_k = eng.get_profile()
        eng.set_profile(0)
    # the drivers' shape of use: ONE video, batch 1, end to end (latency, not part of `value`)
    one = lambda: eng.generate(videos[:1], ts[:1], prompt[:1], heads[:1], n_new, eos=-1, use_graph=True, forced=forced[:1])
    one3 = list(ev_stat(one))                        # (the untimed first call captures the batch-1 decode graph)
    t_one = one3[0]

    # per-rank figures (a collective: every rank gets here): a sub-linear N-GPU curve must show which rank was slow and in which stage
    rank_keys = ["ms_per_step", "vit_slotpool_per_video_ms", "prefill_per_video_ms", "decode_ms_per_step", "weights_load_s", "single_video_latency_ms", "numa_node"]
    per_rank = tdist.gather_floats([dt_local / args.steps * 1e3, t_enc, t_pre, t_dec / (n_new - 1), t_load, t_one, float(place.get("numa_node", -1))])

    if rank == 0:
        vps = world * B * args.steps / dt
        vit_flops = args.frames * (366.0e9 if not args.tiny else 0.0)
        pre_flops = 2 * 6.979e9 * Ls + 32 * 2 * Ls * Ls * 4096 if not args.tiny else 0.0
        k_ms, k_n, k_bytes = prof_k[2], int(prof_k[3]), prof_k[4]
        ach = (k_bytes / (k_ms * 1e-3) / 1e9) if k_ms > 0 else None
        traffic = g_traffic = g_traffic_M = None
        traffic_src = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                # `traffic` is NOT a counter of this run: it is the committed rocprofv3 --pmc pass over the same launch shapes (tools/pmc_kernels.py ->
                # tools/pmc_traffic.py -> profiles/traffic.json, corrected as MI355X_MICROARCH.md prescribes); the line says which file and when it was taken
                traffic_src = {"file": "profiles/traffic.json", "measured": tj.get("measured", "round 4"),
                               "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_kernels.py (same launch shapes as the brackets of this run), FETCH_SIZE x 2 (gfx950)"}
                traffic, g_traffic, g_traffic_M = tj.get("skinny_gateup_bytes_per_launch"), tj.get("gemm_fc1_bytes_per_launch"), tj.get("gemm_fc1_M", 73856)
                if int(prof_k[8]) == 3:      # the wide decode step brackets the decode attention: the committed PMC pass measured it at one batch / context
                    traffic = tj.get("attn_decode_bytes_per_launch") if tj.get("attn_decode_batch") == B else None
                elif int(prof_k[8]) != 1 or args.fp8:
                    traffic = None
            except Exception:
                traffic = g_traffic = None
        g_ms, g_n, g_gf = prof[5], int(prof[6]), prof[7]
        g_M = int(round(g_gf * 1e9 / (2.0 * 4096 * 1024))) if not args.tiny else 0      # rows of the bracketed fc1 launch
        if g_traffic is not None and g_traffic_M != g_M:
            g_traffic = None                                    # the committed PMC pass measured another launch shape
        g_tf = (g_gf / g_ms) if g_ms > 0 else None             # GFLOP / ms = TFLOP/s
        # the persistent kernel the shapes without a residual run on: gemm_w4.hip (4 waves of 128x128) unless TRACE_GEMM_W4=0 sends them back to gemm_pers.hip
        pers_sym = "gemm_w4_kernel" if use_w4 else "gemm_pers_kernel"
        fold_tag = stat_tag = ""                               # (rounds 3-4 tagged the LayerNorm-fold instantiations here; the fold left the product in round 5)
        line = {
            "metric": "videos/sec + decode tok/s, TRACE-7B 128-frame, 1/2/4/8 MI355X",
            "value": vps, "unit": "videos/s", "n_gpus": world, "rccl_ranks": ranks_seen[0], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("fp8 e4m3 weights in the decoder projections: W8A8 (fp8 MFMA, fp32 accumulate) in the prefill GEMMs, " +
                       ("weight-only (bf16 activations, bf16 MFMA) in the decode GEMVs" if args.fp8_scheme == "weight_only" else "W8A8 in the decode GEMVs") +
                       "; bf16 elsewhere (ViT, attention, KV cache, norms, heads)") if args.fp8 else args.dtype), "data": "synthetic",
            "config": {"workload": ("tiny plumbing check" if args.tiny else
                                    f"{preset['name'].replace(' bf16', ' ' + args.dtype)}, {args.frames}x336^2 frames, prefill L={Ls}, {n_new} greedy tokens, heads=[1]"),
                       "baseline_config": args.config,
                       "videos_per_step_per_gpu": B, "frames": args.frames, "prefill_len": Ls, "new_tokens": n_new, "head_schedule": ("forced MR pattern: 14 time + 4 score + 14 text steps per 32 tokens" if preset["schedule"] == "mr" else "forced DVC pattern per event: 14 time + 4 score + 33 text steps") + " (argmax still computed every step)",
                       "decode_launch": ("hipGraph replay (the roofline_hbm bracket is taken in a short eager pass after the timed region)" if args.graph else "eager"), "parallelism": f"dp{world} (replica per GPU)",
                       "step_schedule": ("two-stage pipeline over the timed steps: step k's decode on one HIP stream under step k+1's ViT + prefill on another "
                                         "(fill and drain inside the timed region; the roofline brackets are taken in the fill / drain phases only)"
                                         if pipelined else "steps strictly one after the other"),
                       "weights": "random-init (device RNG), reference architecture"},
            "per_rank": {**{k: [round(r[i], 4) for r in per_rank] for i, k in enumerate(rank_keys)},
                         "ms_per_step_min_median_max": [float(np.min([r[0] for r in per_rank])), float(np.median([r[0] for r in per_rank])), float(np.max([r[0] for r in per_rank]))],
                         "host_placement_rank0": place,
                         "note": "ms_per_step per rank = that rank's own K steps up to its own synchronize (before the closing barrier); `ms_per_step` of the line is the max over ranks with the barrier; stage times are re-timed per rank after the timed region"},
            # what the ids of this configuration are worth (profiles/r04_fp8_error_growth.txt: no e4m3 scheme — W8A8, weight-only, 128-k block-scaled —
            # keeps the 32-layer 13-way time / score decisions within twice the flip count of the reference's own bf16 run; the bf16 / fp16 paths do)
            "parity": ("fp8 tolerance only: logits within the a-priori e4m3 budget of the bf16 path (tests/test_gpu_fp8.py); about a third of the 13-way time / score "
                       "arg-max decisions of a random-weight 32-layer stack differ from the reference's — use the bf16 path where ids must match" if args.fp8 else
                       "reference-anchored: logits inside the reference's own 16-bit-vs-fp32 deviation, 13-way ids bit-exact vs the 16-bit-emulating oracle (tests/test_gpu_*.py)"),
            "steps_repeat_exactly": bool(steps_repeat), **({"steps_repeat_detail": repeat_detail[:12]} if repeat_detail else {}),
            "decode_tok_s": world * B * (n_new - 1) / (t_dec * 1e-3),
            "stages_ms": {"vit_slotpool_per_video": t_enc, "prefill_per_video": t_pre, f"decode_{n_new - 1}_steps_batch{B}": t_dec,
                          "decode_ms_per_step": t_dec / (n_new - 1), "weights_load_s": t_load},
            "single_video_latency_ms": t_one,
            # medians of REPS warm repetitions after the timed region (min / max beside them), and the same two fractions from HIP events INSIDE the
            # timed region: the pipeline's fill batch (sequential runs: every batch), while the encode stage has the GPU to itself
            "mfma_util": {"vit": vit_flops / (t_enc * 1e-3) / 2.5e15, "prefill": pre_flops / (t_pre * 1e-3) / 2.5e15,
                          "vit_min_max": [vit_flops / (enc3[2] * 1e-3) / 2.5e15, vit_flops / (enc3[1] * 1e-3) / 2.5e15],
                          "prefill_min_max": [pre_flops / (pre3[2] * 1e-3) / 2.5e15, pre_flops / (pre3[1] * 1e-3) / 2.5e15],
                          "repetitions": REPS, "how": "algorithmic FLOPs (SURVEY 8d) / median stage time / 2.5e15; each stage warmed once in its call shape, then timed REPS times",
                          "in_timed_region": ({"vit": vit_flops * in_region["videos"] / ((in_region["tower_ms"] + in_region["slotpool_ms"]) * 1e-3) / 2.5e15,
                                               "prefill": pre_flops * in_region["videos"] / (in_region["prefill_ms"] * 1e-3) / 2.5e15,
                                               "videos": in_region["videos"], "tower_ms_per_video": in_region["tower_ms"] / in_region["videos"],
                                               "slotpool_ms_per_video": in_region["slotpool_ms"] / in_region["videos"],
                                               "prefill_ms_per_video": in_region["prefill_ms"] / in_region["videos"],
                                               "how": "HIP-event pairs around the tower / slot-pool / prefill calls of the batch(es) that ran with the GPU to themselves inside the timed region"}
                                              if in_region["videos"] and in_region["prefill_ms"] > 0 and in_region["tower_ms"] > 0 else None)},
            "stages_ms_min_max": {"vit_slotpool_per_video": enc3[1:], "prefill_per_video": pre3[1:], f"decode_{n_new - 1}_steps_batch{B}": dec3[1:], "single_video_latency": one3[1:]},
            # dominant kernel of the run: the 256x256 MFMA GEMM (its four epilogue variants are ~half of GPU time; the probe
            # brackets its largest instance, the ViT fc1 projection, once per video inside the timed region)
            "roofline": {"bound": "mfma", "kernel": f"256x256-tile LDS-DMA MFMA GEMM family (gemm_ldr: 8 MFMA + 4 loader waves, one tile per workgroup; {pers_sym.split('_kernel')[0]}: persistent workgroups): by total time the run's top symbol is gemm_ldr_kernel<EPI_RESIDUAL> (ViT out-proj + fc2, prefill o / down); achieved / frac below are its largest single launch, {pers_sym}<EPI_QUICKGELU{fold_tag}> = ViT fc1 {g_M}x4096x1024 of one {eng.vit_batch_frames if B >= 2 else args.frames}-frame tower call (1 bracketed launch per call); every ViT shape of the family is in `shapes`",
                         "achieved": g_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": (g_tf / 2500.0) if g_tf else None,
                         "traffic": g_traffic, "traffic_source": traffic_src if g_traffic is not None else None, "algorithmic_gflop_per_launch": g_gf, "algorithmic_bytes_per_launch": g_M * 1024 * 2 + 4096 * 1024 * 2 + g_M * 4096 * 2,
                         "avg_launch_ms": g_ms, "samples": g_n,
                         "shapes": {name: {"kernel": sym, "MxNxK": f"{g_M}x{N_}x{K_}", "avg_launch_ms": ms_, "tflops": (gf_ / ms_) if ms_ > 0 else None,
                                           "frac": (gf_ / ms_ / 2500.0) if ms_ > 0 else None}
                                    for name, sym, N_, K_, ms_, gf_ in (
                                        ("vit_qkv", pers_sym + "<EPI_NONE" + fold_tag + ">", 3072, 1024, prof[12], prof[15]),
                                        ("vit_out_proj", "gemm_ldr_kernel<EPI_RESIDUAL>" + stat_tag, 1024, 1024, prof[13], prof[16]),
                                        ("vit_fc1", pers_sym + "<EPI_QUICKGELU" + fold_tag + ">", 4096, 1024, g_ms, g_gf),
                                        ("vit_fc2", "gemm_ldr_kernel<EPI_RESIDUAL>" + stat_tag, 1024, 4096, prof[14], prof[17]))}},
        }
        # dominant HBM-bound kernel of the decode phase
        # whole decode step: algorithmic bytes = the decoder weights once + the heads on the steps that stream them (not counted) + every sequence's KV rows
        kv_step = B * (Ls + n_new / 2.0) * cfg.num_key_value_heads * 128 * 2 * 2 * cfg.num_hidden_layers
        w_step = 2.0 * cfg.num_hidden_layers * (cfg.hidden_size * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * 128 + cfg.hidden_size * cfg.hidden_size
                                                 + 3 * cfg.hidden_size * cfg.intermediate_size) * (0.5 if args.fp8 else 1.0)
        line["decode_step"] = {"algorithmic_gb": (kv_step + w_step) / 1e9, "weights_gb": w_step / 1e9, "kv_gb": kv_step / 1e9,
                               "tb_per_s": (kv_step + w_step) / (t_dec / (n_new - 1) * 1e-3) / 1e12, "gb_per_token": (kv_step + w_step) / B / 1e9}
        bkind = int(prof_k[8])            # which launch the engine put the decode bracket around (it knows which decode path the batch took)
        hbm_kernel = {3: "attn_decode_kernel (decode attention over the batch's KV cache, layer 0, 1 bracketed launch per decode step; wide decode step: "
                         "projections as small-M MFMA GEMMs)",
                      2: "skinny_lds_kernel<EPI_PARTIAL, fused RMSNorm prologue> (batch-1 decode gate|up GEMV, 1 bracketed launch per decode step)",
                      1: ("skinny_fp8_kernel (decode gate|up GEMV on e4m3 weights, 1 bracketed launch per decode step)" if args.fp8 else
                          "skinny_lds_kernel<EPI_PARTIAL,NB,NT=2> (decode gate|up GEMV, 1 bracketed launch per decode step)")}.get(bkind, "none bracketed")
        line["roofline_hbm"] = {"bound": "hbm", "kernel": hbm_kernel,
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None,
                                "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                                "algorithmic_bytes_per_launch": k_bytes, "avg_launch_ms": k_ms, "samples": k_n,
                                "bracket_taken": ("in a short eager pass over the same prefilled KV slots after the timed region (inside it the decode steps are hipGraph replays)"
                                                  if args.graph else "inside the timed region (eager launches; pipelined runs: during the drain only)")}
        if world > 1:
            # timed on rank 0 of the single-GPU run only (the host cores are busy driving N ranks here): the N = 1 line of the same round carries it
            line["cpu_baseline"] = {"value": None, "unit": "videos/s", "cores": None, "kind": "port",
                                    "sample": "not re-timed under N > 1: inherited from the N = 1 run of the same bench.py on the same box (BENCH line)"}
        elif not args.no_cpu_baseline and not args.tiny:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            # cap the thread count: the oracle's small per-head matmuls degrade badly when oversubscribed
            line["cpu_baseline"] = cpu_baseline(cfg, n_new, max(1, min(32, cores)))
        elif args.tiny:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    eng.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
