"""Kernel microbenchmarks at TRACE-7B shapes (run on the GPU box): prints achieved TFLOP/s / GB/s per kernel.
python tools/microbench.py [--quick]"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E  # noqa: E402
from trace_amd.engine import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


res = []
quick = "--quick" in sys.argv
variants = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--variant=")] or [0]
only_gemm = "--gemm-only" in sys.argv
only_attn = "--attn-only" in sys.argv
# --- prefill-shaped GEMMs (ViT M = 128*577, LLM M = 1968)
for name, M, N, K, epi in [
    ("vit_qkv", 73856, 3072, 1024, E.EPI_NONE), ("vit_out", 73856, 1024, 1024, E.EPI_RESIDUAL),
    ("vit_fc1", 73856, 4096, 1024, E.EPI_QUICKGELU), ("vit_fc2", 73856, 1024, 4096, E.EPI_RESIDUAL),
    ("llm_qkv", 1968, 6144, 4096, E.EPI_NONE), ("llm_gateup", 1968, 28672, 4096, E.EPI_SWIGLU),
    ("llm_down", 1968, 4096, 14336, E.EPI_RESIDUAL), ("sq4096", 4096, 4096, 4096, E.EPI_NONE),
]:
    if only_attn:
        break
    if quick and M > 8192:
        M = 8192
    A, W = rnd(M, K), rnd(N, K, scale=0.02)
    b = rnd(N) if "vit" in name else None
    No = N // 2 if epi == E.EPI_SWIGLU else N
    R = rnd(M, No) if epi == E.EPI_RESIDUAL else None
    for v in variants:
        ops.set_gemm_variant(v)
        ms = timeit(lambda: ops.gemm(A, W, bias=b, R=R, epilogue=epi), iters=5)
        tf = 2.0 * M * N * K / ms / 1e9
        res.append({"kernel": "gemm_" + name, "variant": v, "M": M, "N": N, "K": K, "ms": round(ms, 4), "TFLOPs": round(tf, 1), "mfma_frac": round(tf / 2500, 3)})
        print(res[-1], flush=True)
    ops.set_gemm_variant(0)
    del A, W, R
if only_gemm:
    sys.exit(0)
# --- attention
for name, Bn, n, heads, kvh, hd, causal in [("attn_vit", 32 if quick else 170, 577, 16, 16, 64, False), ("attn_prefill", 2, 1967, 32, 8, 128, True),
                                            ("attn_prefill_c5", 1, 3834, 32, 8, 128, True)]:
    q, k, v = rnd(Bn, n, heads, hd), rnd(Bn, n, kvh, hd), rnd(Bn, n, kvh, hd)
    ms = timeit(lambda: ops.attention(q, k, v, causal, 1 / math.sqrt(hd)), iters=5)      # includes the V transpose kernel
    fl = 4.0 * Bn * heads * n * n * hd * (0.5 if causal else 1.0)
    res.append({"kernel": name, "ms": ms, "TFLOPs": fl / ms / 1e9})
    print(res[-1], flush=True)
if only_attn:
    sys.exit(0)
# --- decode weight streaming
for name, N, K, epi in [("dec_qkv", 6144, 4096, E.EPI_NONE), ("dec_o", 4096, 4096, E.EPI_RESIDUAL),
                        ("dec_gateup", 28672, 4096, E.EPI_SWIGLU), ("dec_down", 4096, 14336, E.EPI_RESIDUAL),
                        ("dec_lmhead", 32000, 4096, E.EPI_NONE)]:
    for Bn in (1, 8, 16):
        # rotate over several weight copies so the stream comes from HBM, not the 256 MB Infinity Cache
        copies = max(2, int(600e6 // (N * K * 2)))
        Ws = [rnd(N, K, scale=0.02) for _ in range(copies)]
        X = rnd(Bn, K)
        No = N // 2 if epi == E.EPI_SWIGLU else N
        R = rnd(Bn, No) if epi == E.EPI_RESIDUAL else None
        st = {"i": 0}

        def run():
            st["i"] = (st["i"] + 1) % copies
            ops.skinny_gemm(X, Ws[st["i"]], R=R, epilogue=epi)
        ms = timeit(run, iters=4 * copies, warmup=copies)
        gbs = N * K * 2 / ms / 1e6
        res.append({"kernel": "skinny_" + name, "B": Bn, "N": N, "K": K, "us": ms * 1e3, "GBps": gbs, "hbm_frac": gbs / 8000})
        print(res[-1], flush=True)
        del Ws
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)
