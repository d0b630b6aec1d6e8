"""Decode GEMV microbenchmark at the TRACE-7B shapes: us per launch and weight-stream TB/s for B = 1..32.
Rotates over several weight copies so that neither L2 nor the 256 MB infinity cache can hold the stream.
--dbg=N sets the kernel's microbenchmark mode (0 product path; 3 stop before the epilogue); --rowmajor skips the tile layout."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
dbgs = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--dbg=")] or [0]
Bs = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--B=")] or [1, 16, 32]
tiled = "--rowmajor" not in sys.argv
shapes = [("qkv", 6144, 4096, E.EPI_NONE), ("o", 4096, 4096, E.EPI_PARTIAL), ("gate|up", 28672, 4096, E.EPI_SWIGLU),
          ("down", 4096, 14336, E.EPI_PARTIAL)]
tot = {}
for name, N, K, epi in shapes:
    ncopy = max(2, int(1.2e9 // (N * K * 2)))
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
    if tiled:
        Ws = [ops.tile_pack(W) for W in Ws]
    for B in Bs:
        X = torch.randn(B, K, device=dev).to(torch.bfloat16)
        for dbg in dbgs:
            ops.set_gemm_variant(200 + dbg)
            for W in Ws: ops.skinny_gemm(X, W, epilogue=epi, tiled=tiled, want_partial=False)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 4
            s.record()
            for _ in range(iters):
                for W in Ws: ops.skinny_gemm(X, W, epilogue=epi, tiled=tiled, want_partial=False)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / (iters * ncopy) * 1e3
            tot[(B, dbg)] = tot.get((B, dbg), 0) + us
            print(f"{name:8s} N={N} K={K} B={B:2d} dbg={dbg} ks={ops.skinny_ks(N, K, epi, B)}: {us:6.1f} us  {N * K * 2 / us / 1e6:.2f} TB/s", flush=True)
        ops.set_gemm_variant(200)
    del Ws
for k, v in sorted(tot.items()):
    print(f"sum of the four GEMVs  B={k[0]:2d} dbg={k[1]}: {v:.1f} us/layer")
