"""Small fixed workload for rocprofv3 --pmc passes: ViT attention, ViT fc1 GEMM, decode gate|up GEMV (batch 64)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
what = sys.argv[1:] or ["all"]
if "all" in what or "attn" in what:
    q, k, v = rnd(32, 577, 16, 64), rnd(32, 577, 16, 64), rnd(32, 577, 16, 64)
    for _ in range(3):
        ops.attention(q, k, v, False, 0.125)
if "all" in what or "gemm" in what:
    A, W, b = rnd(170 * 577, 1024), rnd(4096, 1024, scale=0.02), rnd(4096)      # the bench's probe shape: one 170-frame ViT call
    for _ in range(3):
        ops.gemm(A, W, bias=b, epilogue=E.EPI_QUICKGELU)
if "all" in what or "gemv" in what:      # the decode step's dominant kernel: gate|up GEMV, batch 64 (the bench default), tile-layout weights, fp32 partial rows
    Ws = [ops.tile_pack(rnd(28672, 4096, scale=0.02)) for _ in range(3)]
    X = rnd(64, 4096)
    for i in range(6):
        ops.skinny_gemm(X, Ws[i % 3], epilogue=E.EPI_PARTIAL, tiled=True, want_partial=False)
torch.cuda.synchronize()
