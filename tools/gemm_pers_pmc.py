"""Workload for rocprofv3 --pmc passes over the 256x256 GEMM kernels: gemm_ldr (variant 4) and gemm_pers (5) on one shape.
python tools/gemm_pers_pmc.py M N K [epilogue]      (default: the long-K shape 3934 x 4096 x 14336, one tile per workgroup)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
M, N, K = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (3934, 4096, 14336)
epi = int(sys.argv[4]) if len(sys.argv) >= 5 else E.EPI_NONE
A, W = rnd(M, K), rnd(N, K, scale=0.02)
R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
for v in (4, 5):
    ops.set_gemm_variant(v)
    for _ in range(3):
        ops.gemm(A, W, R=R, epilogue=epi)
ops.set_gemm_variant(0)
torch.cuda.synchronize()
