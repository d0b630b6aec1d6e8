"""How do the MFMA GEMM kernels do in the decode regime (M = 64 / 128 rows, weights streamed once from HBM)?  Times ops.gemm on the four
decoder projection shapes beside the decode GEMV (skinny_lds, EPI_PARTIAL + its combine kernels are not included: GEMV time only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.engine import ops, EPI_NONE, EPI_SWIGLU, EPI_PARTIAL

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # rotate through several weight copies so that no launch finds its weights in the Infinity Cache
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

shapes = [("gate|up", 28672, 4096, EPI_SWIGLU), ("down", 4096, 14336, EPI_NONE), ("qkv", 6144, 4096, EPI_NONE), ("o", 4096, 4096, EPI_NONE)]
NW = 6
for name, N, K, epi in shapes:
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(NW)]
    Wt = [ops.tile_pack(w) for w in Ws]
    for M in (64, 128):
        X = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        res = {}
        for var in (0, 2):
            ops.set_gemm_variant(var)
            try:
                res[f"gemm v{var}"] = timeit(lambda i=0: ops.gemm(X, Ws[i % NW], epilogue=epi))
            except Exception as e:
                res[f"gemm v{var}"] = f"err {e}"
        ops.set_gemm_variant(0)
        if M <= 64:
            res["skinny partial"] = timeit(lambda i=0: ops.skinny_gemm(X, Wt[i % NW], epilogue=EPI_PARTIAL, tiled=True, want_partial=False))
        mb = N * K * 2 / 1e6
        print(f"{name:8s} N={N:6d} K={K:6d} M={M:4d} weights {mb:6.1f} MB | " + " | ".join(
            f"{k}: {v:7.1f} us = {mb / v * 1e6 / 1e6:5.2f} TB/s" if isinstance(v, float) else f"{k}: {v}" for k, v in res.items()), flush=True)
    del Ws, Wt
